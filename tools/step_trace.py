"""One training step as an ordered kernel list from a rocprofv3 --kernel-trace CSV: python tools/step_trace.py <kernel_trace.csv> [out.txt]
The step = the dispatches between the last two groups of fused-AdamW launches.  Prints name, duration, gap to the previous kernel,
and a summary by kernel name."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n.split("(")[0][:100]
is_opt = [("FusedOptimizer" in r["Kernel_Name"]) for r in rows]
ends = [i for i in range(len(rows)) if is_opt[i] and (i + 1 == len(rows) or not is_opt[i + 1])]
assert len(ends) >= 2, "need two optimiser steps in the trace"
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
t_prev = int(step[0]["Start_Timestamp"])
tot = 0
by = collections.OrderedDict()
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    tot += d
    nm = short(r["Kernel_Name"])
    print(f"{d:9.1f} us  gap {max(0, (s - t_prev)) / 1e3:7.1f}  {nm}", file=out)
    t_prev = e
    k = by.setdefault(nm, [0, 0.0])
    k[0] += 1; k[1] += d
wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
print(f"\n# {len(step)} launches, kernel time {tot/1e3:.3f} ms, wall {wall/1e3:.3f} ms", file=out)
for nm, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"# {c:4d} x {d/c:8.1f} us = {d/1e3:7.3f} ms  {nm}", file=out)
