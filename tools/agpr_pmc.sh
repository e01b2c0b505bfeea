# PMC counters of the 8 x 256 bf16 kernel in both forms (A: 8 waves x 1 block, B: 4 waves x 2 blocks, AGPR).  usage: bash tools/agpr_pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-agpr_pmc}; mkdir -p $O
for F in 0 1; do
  export MODA_MLP_AGPR=$F
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/f${F}_1 -o p -- python3 tools/mlp_one.py 3 > $O/f${F}_1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC --output-format csv -d $O/f${F}_2 -o p -- python3 tools/mlp_one.py 3 > $O/f${F}_2.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/f${F}_k -o k -- python3 tools/mlp_one.py 6 > $O/f${F}_k.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for F in (0, 1):
    agg = collections.defaultdict(list)
    for f in glob.glob("$O/f%d_[12]/**/*counter_collection.csv" % F, recursive=True):
        for r in csv.DictReader(open(f)):
            if "mlp_fused_kernel<256" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob("$O/f%d_k/**/*kernel_trace.csv" % F, recursive=True):
        for r in csv.DictReader(open(f)):
            if "mlp_fused_kernel<256" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    print("form", "B (4x2 AGPR)" if F else "A (8x1)", "kernel ms", ["%.3f" % d for d in dur])
    for k in sorted(m): print("   %-28s %.4g" % (k, m[k]))
    if "GRBM_GUI_ACTIVE" in m and dur:
        clk = m["GRBM_GUI_ACTIVE"] / 8 / (sum(dur[1:]) / max(len(dur) - 1, 1) * 1e-3) / 1e9
        print("   effective clock %.3f GHz; MFMA pipe busy %.1f %%" % (clk, 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
