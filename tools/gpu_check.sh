cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/check; mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
( time python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -4 $O/bench.err
python - <<'P'
import json
j=json.loads(open('gpurun_out/check/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['roofline']['frac'])
c=j['configs']['cfg5_ama_fine128+128_cse']; print({k:(round(v,1) if isinstance(v,float) else v) for k,v in c.items() if k not in('how',)})
P
