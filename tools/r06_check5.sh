cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_fp16.py tests/test_gpu_combined_configs.py tests/test_gpu_early_term.py tests/test_gpu_default_mode.py -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
for P in bf16 fp16; do python tools/cfg_bench.py cfg5 65536 $P 2>/dev/null | tail -1; MODA_REUSE_COARSE=0 python tools/cfg_bench.py cfg5 65536 $P 2>/dev/null | tail -1; done
python tools/fp16_cfg5_probe.py 2>&1 | grep -v amdgpu.ids | head -3
