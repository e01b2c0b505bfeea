# PMC passes over tools/gemm_x3_bench.py; usage: bash tools/x3_pmc.sh <tag> [mode]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-x3pmc}; MODE=${2:-bf16x6}
O=gpurun_out/$T
mkdir -p $O
timeout 120 python3 tools/gemm_x3_bench.py $MODE > $O/times.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p -- python3 tools/gemm_x3_bench.py $MODE 262144 256 4 > $O/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -o p -- python3 tools/gemm_x3_bench.py $MODE 262144 256 4 > $O/p2.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_x3" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O + "/pmc.txt", "w") as fo:
    for k, cs in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(cs.items()):
            fo.write(f"   {c:28s} {sum(v)/len(v):16.0f}  ({len(v)} dispatches)\n")
PY
rm -rf $O/p1 $O/p2
cat $O/times.txt
