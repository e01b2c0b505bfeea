#!/bin/bash
# usage (GPU box): bash tools/train_stats.sh <tag> [precision]  -- per-step kernel time by kernel of the eagerly launched cfg4 training step
O=$GRAFT_REPO_ROOT/gpurun_out/$1
P=${2:-bf16}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kst -o train -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --precision $P --no-graph --settle-steps 2 --steps 10 --warmup 0 > $O/ks_train.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, re
f = glob.glob("gpurun_out/$1/kst/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 12.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
print("kernel time per step: %.0f us over %d kernels" % (tot, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:26]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    print("%7.1f us/step %5.1f calls/step avg %6.1f us  %s" % (float(r["TotalDurationNs"]) / steps / 1e3, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, n[:100]))
PY
