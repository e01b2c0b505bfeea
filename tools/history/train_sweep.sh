# usage (on the GPU box): bash tools/train_sweep.sh VAR v1 v2 ...   -> training step time for each value of the env var
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for t in "$@"; do
  env $VAR=$t python bench.py --mode train --precision ${PREC:-bf16} --steps 30 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],3), d['loss'])" "$VAR=$t"
done
