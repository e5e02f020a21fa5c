"""Summaries out of rocprofv3's rocpd databases (the default output format of this ROCm):
   db_summary.py stats <db>   per-kernel calls / average / total (the --stats table)
   db_summary.py pmc <db>...  mean counter value per dispatch and kernel"""
import re, sqlite3, sys
def short(n):
    m = re.match(r'(?:void )?(svk::\w+(?:<[^>]*>)?)', n)
    return m.group(1) if m else n[:48]
mode = sys.argv[1]
for path in sys.argv[2:]:
    c = sqlite3.connect(path)
    print("#", path)
    if mode == "stats":
        print("Name,Calls,TotalDurationNs,AverageNs,Percentage")
        for n, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print('"%s",%d,%d,%.1f,%.2f' % (short(n), calls, round(tot * 1e3), avg * 1e3, pct))
    else:
        q = "select kernel_name,counter_name,count(*),avg(value) from counters_collection group by kernel_name,counter_name order by kernel_name"
        for n, cn, k, v in c.execute(q):
            if "svk::" in n:
                print("%-34s %-12s n=%3d mean=%14.1f" % (short(n), cn, k, v))
