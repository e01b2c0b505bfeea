cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
for i in 1 2 3; do ( time python -m pytest tests -m gpu -q -x ) > $O/pytest_$i.log 2>&1; echo "run $i rc=$?"; tail -3 $O/pytest_$i.log | head -1; done
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'P'
import json
j=json.loads(open('gpurun_out/r06f/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['dtype'], j['roofline']['frac'], j['roofline']['traffic'], j['roofline']['traffic_source']['fresh'])
for k,v in j['configs'].items():
    if 'roofline_hbm' in v: print(k, v['ms_per_step'], v['roofline']['frac'], (v['roofline_hbm'] or {}).get('frac'), (v['roofline_hbm'] or {}).get('profile_is_of_this_build'))
print(j['cpu_baseline']['value'], j['cpu_baseline']['sample'][:80])
P
