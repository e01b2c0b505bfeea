cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g
python bench.py > gpurun_out/g/bench.json 2> gpurun_out/g/bench.err
python bench.py --mode train > gpurun_out/g/bench_train.json 2>> gpurun_out/g/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/g/ks -o fwd -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/g/ks_fwd.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/g/kst -o train -- python3 bench.py --mode train --steps 5 --warmup 2 > gpurun_out/g/ks_train.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/g/pmc_fetch -o p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/g/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/g/pmc_write -o p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/g/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/g/pmc_sq1 -o p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/g/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/g/pmc_sq2 -o p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/g/pmc_sq2.log 2>&1
ls gpurun_out/g/*; cat gpurun_out/g/bench.json | cut -c1-300
