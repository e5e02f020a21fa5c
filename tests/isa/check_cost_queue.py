"""Checks a device listing (hipcc --cuda-device-only -S) for the one thing the hand-counted memory queue of k_cost2 cannot survive:
an instruction outside the inline assembly that writes a queue register, or a move that reads one (a copy of a register whose load
may still be in flight).  check_cost_queue.py [listing]; exit status 1 and the offending lines if any."""
import re, sys

def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()

def check(path):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*k_cost2\w*:", l)]
    bad, seen = [], 0
    for s in starts:
        e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
        name = lines[s].split(":")[0]
        # regions: from the first inline-assembly load (a radius case primes its queue) to the drain behind that case's loop
        regions, in_asm, first, q = [], False, None, set()
        for i in range(s, e):
            l = lines[i]
            if "#ASMSTART" in l: in_asm = True
            elif "#ASMEND" in l: in_asm = False
            elif in_asm:
                m = re.match(r"\s+global_load_dword(?:x2)?\s+([^,]+),", l)
                if m:
                    q |= regs(m.group(1).strip()); first = first if first is not None else i
                if "vmcnt(0)" in l and first is not None:
                    regions.append((first, i, q)); first, q = None, set()
        if regions: seen += 1
        for first, last, _ in regions:
            in_asm, q = True, set()    # (first = the region's first load, inside its assembly block)
            for i in range(first, last):
                l = lines[i]
                if "#ASMSTART" in l: in_asm = True; continue
                if "#ASMEND" in l: in_asm = False; continue
                m = re.match(r"\s+([a-z][a-z0-9_]+)\s+(.*)", l)
                if not m or l.lstrip().startswith((";", ".")): continue
                op, args = m.group(1), [a.strip() for a in m.group(2).split(";")[0].split(",")]
                if in_asm:
                    if op.startswith("global_load"): q |= regs(args[0])   # a queue register from its first load on
                    continue
                stores = op.startswith(("global_store", "ds_write", "scratch_store", "buffer_store", "s_"))
                if not stores and args and regs(args[0]) & q:
                    bad.append((name, i + 1, "writes a queue register", l.strip()))
                if op.startswith(("v_mov", "v_accvgpr", "scratch_store")) and any(regs(a) & q for a in args[1:] + (args[:1] if op.startswith("scratch_store") else [])):
                    bad.append((name, i + 1, "copies a queue register", l.strip()))
    return seen, bad

if __name__ == "__main__":
    seen, bad = check(sys.argv[1] if len(sys.argv) > 1 else "/tmp/sv.s")
    print("%d kernels with a hand-counted queue, %d offending instructions" % (seen, len(bad)))
    for b in bad: print("  %s:%d %s: %s" % (b[0][:48], b[1], b[2], b[3]))
    sys.exit(1 if bad or not seen else 0)
