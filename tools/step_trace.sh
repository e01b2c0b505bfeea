# ordered kernel trace of one training step (eager launches): bash tools/step_trace.sh <outdir> <bf16|fp32> [name] [extra bench.py args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-trace}; P=${2:-bf16}; NAME=${3:-$P}; shift; shift; shift
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt_$NAME -o train -- python3 bench.py --mode train --precision $P --no-graph --settle-steps 4 --steps 3 --warmup 1 "$@" > $O/kt_$NAME.log 2>&1
python tools/step_trace.py $(find $O/kt_$NAME -name "*kernel_trace.csv" | head -1) $O/step_$NAME.txt
tail -60 $O/step_$NAME.txt
