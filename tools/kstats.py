"""Print a rocprofv3 kernel_stats.csv compactly: name (shortened), calls, total ms, avg us, %."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:top]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"^void ", "", n)
    n = n.split("(")[0][:90]
    print(f"{n:90s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} ms {float(r['AverageNs'])/1e3:10.1f} us {float(r['Percentage']):6.2f}%")
