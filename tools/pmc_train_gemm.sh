# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the training step's GEMM kernels; usage: bash tools/pmc_train_gemm.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/${1:-pg}
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/$c -o p -- python3 bench.py --mode train --precision bf16 --no-graph --settle-steps 1 --steps 2 --warmup 1 > $O/log_$c.txt 2>&1
done
python3 - $O <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k and "gemv" not in k:
            continue
        key = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O + "/gemm_traffic.txt", "w") as fo:
    for k, cs in sorted(agg.items()):
        f = cs.get("FETCH_SIZE", [0]); w = cs.get("WRITE_SIZE", [0])
        fo.write(f"{k:72s} n={len(f):4d} fetch(raw KB->MB) {sum(f)/len(f)/1024:9.1f}  x2 {2*sum(f)/len(f)/1024:9.1f}  write MB {sum(w)/len(w)/1024:9.1f}\n")
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
cat $O/gemm_traffic.txt
