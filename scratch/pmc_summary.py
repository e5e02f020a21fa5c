"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch."""
import csv, sys, collections, re
def short(name):
    m = re.match(r'(?:void )?(svk::\w+(?:<[^>]*>)?)', name)
    return m.group(1) if m else name[:40]
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    print("#", path)
    for (k, c), (s, n) in sorted(acc.items()):
        if k.startswith("svk::"):
            print("%-34s %-12s n=%3d mean=%14.1f" % (k, c, n, s / n))
