"""Instruction mix of selected kernels in a device-only assembly listing
   (hipcc --cuda-device-only -S ...): isa_mix.py file.s substr [substr...]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for k, (i, name) in enumerate(starts):
    if not any(s in name for s in sys.argv[2:]):
        continue
    end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    ops = collections.Counter()
    for l in lines[i:end]:
        m = re.match(r"^\s+([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+|flat_[a-z0-9_]+)", l)
        if m: ops[m.group(1)] += 1
    tot = sum(ops.values()); valu = sum(v for k2, v in ops.items() if k2.startswith("v_"))
    print("%s\n  total %d valu %d cndmask %d  top: %s" % (name[:90], tot, valu, ops["v_cndmask_b32"] + ops["v_cndmask_b32_e32"] + ops["v_cndmask_b32_e64"],
          ", ".join("%s %d" % kv for kv in ops.most_common(16))))
