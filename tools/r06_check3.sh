cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
python tools/gemv_ab.py 2048 > $O/gemv_ab.txt 2>&1; cat $O/gemv_ab.txt | grep -v amdgpu.ids
( time python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
for a in "" "--rays 8192 --samples 256"; do python bench.py --mode train --precision bf16 --steps 50 --settle-steps 20 $a 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['loss'], j['roofline']['frac'])"; done
bash tools/step_trace.sh r06d bf16 cfg4_2048x128 > /dev/null 2>&1; grep "^# " $O/step_cfg4_2048x128.txt | head -3
