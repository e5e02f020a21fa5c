"""Instruction counts over line ranges of an assembly listing: isa_range.py file.s name:a-b ..."""
import re, sys, collections
L = open(sys.argv[1]).read().split("\n")
tot = collections.Counter()
for spec in sys.argv[2:]:
    name, rg = spec.split(":"); a, b = map(int, rg.split("-"))
    c = collections.Counter()
    for l in L[a - 1:b]:
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\s", l + " ")
        if not m or l.strip().startswith("."): continue
        o = m.group(1)
        k = "mov" if o.startswith("v_mov") else "valu" if o.startswith("v_") else "nop" if o == "s_nop" else "salu" if o.startswith("s_") else "ds" if o.startswith("ds_") else "vmem"
        c[k] += 1
        if k == "mov": c["valu"] += 1
    tot.update(c)
    print("%-14s" % name, dict(c))
print("%-14s" % "total", dict(tot))
