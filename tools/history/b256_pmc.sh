#!/bin/bash
# usage (GPU box): bash tools/b256_pmc.sh <tag> [lib]  -- LDS / wait counters of bwd256_kernel (tools/bwd256_probe.py 262144)
O=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $O
[ -n "$2" ] && export MODA_LIB_PATH=$GRAFT_REPO_ROOT/$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -io "SQ_[A-Z_]*LDS[A-Z_]*" | sort -u | tr '\n' ' ' > $O/lds_counters.txt
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -o p -- python3 $GRAFT_REPO_ROOT/tools/bwd256_probe.py 262144 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -o p -- python3 $GRAFT_REPO_ROOT/tools/bwd256_probe.py 262144 > $O/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
for d in ("p1", "p2"):
    agg = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/$1/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "bwd256" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(d, k, "mean per launch %.4g over %d" % (sum(v) / len(v), len(v)))
PY
cat $O/lds_counters.txt; tail -n 3 $O/p1.log
