cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
python tools/glue_shapes.py > $O/glue.txt 2>&1; head -50 $O/glue.txt
