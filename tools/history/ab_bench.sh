# usage (GPU box): bash tools/ab_bench.sh "<flags A>" "<flags B>" ...   -> rebuilds the library with each flag set and runs bench.py (A/B on ONE box)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in "$@"; do
  MODA_HIPCC_FLAGS="$v" python -c "from moda_amd import build; build.build(force=True, verbose=False)" > /dev/null 2>&1
  python bench.py --steps ${STEPS:-40} --warmup ${WARM:-5} --no-cpu-baseline --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(repr(sys.argv[1]), round(d['ms_per_step'],3), round(d['roofline']['ms_per_launch'],3), d['roofline']['other_kernels_ms_per_launch'])" "$v"
done
done
