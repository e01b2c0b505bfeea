# ordered kernel trace of one training step (eager launches): bash tools/step_trace.sh <outdir> <bf16|fp32>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-trace}; P=${2:-bf16}
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt_$P -o train -- python3 bench.py --mode train --precision $P --no-graph --settle-steps 4 --steps 3 --warmup 1 > $O/kt_$P.log 2>&1
python tools/step_trace.py $(find $O/kt_$P -name "*kernel_trace.csv" | head -1) $O/step_$P.txt
tail -60 $O/step_$P.txt
