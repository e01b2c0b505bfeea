# usage (GPU box): bash tools/ab_flags.sh "<flags A>" "<flags B>" ...  (flags may start with '-'): rebuild + bench, two rounds
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in "$@"; do
  MODA_HIPCC_FLAGS="$v" python -c "from moda_amd import build; build.build(force=True, verbose=False)" > /dev/null 2>&1
  python bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-fp32 > /tmp/ab_line.json 2>/dev/null
  python - "$v" <<'PY'
import sys, json
try:
    d = json.loads(open('/tmp/ab_line.json').read())
    print('[%s]' % sys.argv[1], round(d['ms_per_step'], 3), round(d['roofline']['ms_per_launch'], 3), d['roofline']['other_kernels_ms_per_launch'])
except Exception as e:
    print('[%s] failed: %s' % (sys.argv[1], e))
PY
done
done
