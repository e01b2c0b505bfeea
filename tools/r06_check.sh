# full GPU suite + smoke + default bench (timed)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
( time python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -5 $O/bench.err; cut -c1-600 $O/bench.json
