#!/bin/bash
# usage (GPU box): bash tools/b256_prof.sh <tag>   -- per-kernel times of tools/bwd256_ab.py (fused vs two-launch hidden layers)
O=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o p -- python3 $GRAFT_REPO_ROOT/tools/bwd256_ab.py 2 > $O/run.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/$1/ks/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Name"][:110])
PY
tail -n 2 $O/run.log
