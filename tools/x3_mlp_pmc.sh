# PMC of the fused 8 x 256 split-bf16 kernel (the hierarchical pre-pass's network in the fp16 mode): tools/mlp_one.py with MODA_ONE_PREC=bf16x3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/x3mlp; mkdir -p $O
export MODA_ONE_PREC=${1:-bf16x3}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o r -- python3 tools/mlp_one.py 4 > $O/ks.log 2>&1
python tools/kstats.py $(find $O/ks -name "*kernel_stats.csv" | head -1) 3
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d $O/p1 -o p -- python3 tools/mlp_one.py 3 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p2 -o p -- python3 tools/mlp_one.py 3 > $O/p2.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mlp_fused_kernel<256" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(anonymous namespace)::")[-1][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}  ({len(v)} dispatches)")
PY
