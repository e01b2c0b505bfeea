# PMC counters of the training step's kernels (eager, few steps); usage: bash tools/pmc_train.sh "<counters>" <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/$2
mkdir -p $O
rocprofv3 --pmc $1 --output-format csv -d $O -o p -- python3 bench.py --mode train --precision bf16 --no-graph --settle 0 --steps 2 --warmup 1 > $O/log.txt 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$O/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "mlp_fused_kernel" not in k: continue
    key=k[k.index("mlp_fused_kernel"):][:70]
    agg[key][r["Counter_Name"]]+=float(r["Counter_Value"])
    cnt[(key,r["Counter_Name"])]+=1
for k,v in agg.items():
    print(k)
    for c,x in v.items(): print("   %-28s %14.0f per dispatch (%d dispatches)"%(c, x/cnt[(k,c)], cnt[(k,c)]))
PY
