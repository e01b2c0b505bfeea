# HBM bytes of ONE training step, summed over every kernel: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE -- separate
# passes, as the guide prescribes) of the eagerly launched step, 2 x FETCH + WRITE per the gfx950 correction; result into
# profiles/traffic.json["train_step_<precision><suffix>"] with the kernel-source stamp.
# usage: bash tools/pmc_train_step.sh <tag> [precision] [suffix] [what] [extra bench.py args...]
#   e.g. bash tools/pmc_train_step.sh r06 bf16 _cfg4_8192x256 "cfg4 training step (8192 rays x 256 samples)" --rays 8192 --samples 256
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
T=${1:-r06}; P=${2:-bf16}; SUF=${3:-}; WHAT=${4:-cfg4 training step (2048 rays x 128 samples)}
shift; shift; shift; shift
O=gpurun_out/$T/pmc_train_$P$SUF
mkdir -p $O
STEPS=4; SETTLE=2
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $O/$C -o p -- python3 bench.py --mode train --precision $P --no-graph --settle-steps $SETTLE --steps $STEPS --warmup 0 "$@" > $O/$C.log 2>&1
done
python3 tools/pmc_train_total.py "$P:$SUF:$WHAT" $((STEPS + SETTLE)) $(find $O -name "*counter_collection.csv") | tee $O/summary.txt
