# kernel stats of the bf16 training step (eager), top lines matching a pattern; usage: bash tools/train_kstat.sh <pattern> [n]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o train -- python3 bench.py --mode train --precision bf16 --no-graph --settle-steps 2 --steps 10 --warmup 0 > gpurun_out/kt.log 2>&1
F=$(find gpurun_out/kt -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python tools/kstats.py $F ${2:-60} | grep -i "total kernel\|$1"
rm -rf gpurun_out/kt
