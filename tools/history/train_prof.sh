# kernel stats of the training step (eager, bf16 mode) with the bf16 store on / off; usage: bash tools/train_prof.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-tp}
O=gpurun_out/$T
mkdir -p $O
for s in 1 0; do
  export MODA_TRAIN_BF16_STORE=$s
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$s -o train -- python3 bench.py --mode train --precision bf16 --no-graph --settle 0 --steps 10 --warmup 2 > $O/ks_train$s.log 2>&1
  cp $(find $O/ks$s -name "*kernel_stats.csv" | head -1) $O/train_store${s}_kernel_stats.csv
  python tools/kstats.py $O/train_store${s}_kernel_stats.csv 16
done
