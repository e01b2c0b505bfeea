cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
