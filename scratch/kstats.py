import csv, sys, re
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r["Name"]
        m = re.match(r'(?:void )?(svk::\w+(?:<[^>]*>)?)', n)
        print("%-32s calls=%4s avg_us=%9.1f total_us=%10.1f %5s%%" % (m.group(1) if m else n[:30], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3, r["Percentage"]))
