# usage (GPU box): [MODE=train] bash tools/ab_flags.sh "<flags A>" "<flags B>" ...  (flags may start with '-'): rebuild + bench, two rounds
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in "$@"; do
  MODA_HIPCC_FLAGS="$v" python -c "from moda_amd import build; build.build(force=True, verbose=False)" > /dev/null 2>&1
  if [ "${MODE:-render}" = "train" ]; then python bench.py --mode train --precision bf16 --steps ${STEPS:-60} > /tmp/ab_line.json 2>/dev/null
  else python bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-fp32 > /tmp/ab_line.json 2>/dev/null; fi
  python - "$v" <<'PY'
import sys, json
try:
    d = json.loads(open('/tmp/ab_line.json').read())
    r = d.get('roofline') or {}
    print('[%s]' % sys.argv[1], round(d['ms_per_step'], 3), round(r.get('ms_per_launch', 0.0), 3), r.get('other_kernels_ms_per_launch'))
except Exception as e:
    print('[%s] failed: %s' % (sys.argv[1], e))
PY
done
done
