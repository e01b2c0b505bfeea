# kernel stats of the training step in the bf16x6 (fast parity) mode; usage: bash tools/x6_prof.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-x6p}
O=gpurun_out/$T
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o train -- python3 bench.py --mode train --precision bf16x6 --no-graph --settle-steps 2 --steps 10 --warmup 0 > $O/ks_train.log 2>&1
F=$(find $O/ks -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cp "$F" $O/train_bf16x6_kernel_stats.csv; python tools/kstats.py $O/train_bf16x6_kernel_stats.csv 40 > $O/summary.txt; fi
rm -rf $O/ks
tail -3 $O/ks_train.log
