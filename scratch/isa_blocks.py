"""Per-basic-block instruction mix of one kernel in a device-only assembly listing:
   isa_blocks.py file.s <kernel substr> [min_instrs]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for k, (i, name) in enumerate(starts):
    if sub not in name: continue
    end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    print(name[:100])
    blocks = []; cur = ("entry", [])
    for l in lines[i + 1:end]:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            blocks.append(cur); cur = (m.group(1), [])
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\s", l + " ")
        if m and not l.strip().startswith("."): cur[1].append(m.group(1))
    blocks.append(cur)
    for nm, ops in blocks:
        if len(ops) < mn: continue
        c = collections.Counter(ops)
        valu = sum(v for o, v in c.items() if o.startswith("v_"))
        pk = sum(v for o, v in c.items() if o.startswith("v_pk_"))
        dppn = 0
        mov = sum(v for o, v in c.items() if o.startswith("v_mov") or o.startswith("v_accvgpr"))
        print("  %-14s n=%4d valu=%4d pk=%3d mov=%3d nop=%3d salu=%3d ds=%3d vmem=%3d | %s" % (nm, len(ops), valu, pk, mov, c["s_nop"],
              sum(v for o, v in c.items() if o.startswith("s_") and o != "s_nop"), sum(v for o, v in c.items() if o.startswith("ds_")),
              sum(v for o, v in c.items() if o.startswith("global_") or o.startswith("buffer_")),
              ", ".join("%s %d" % kv for kv in c.most_common(12))))
