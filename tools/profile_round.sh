# Round profile of bench.py on one MI355X (run through gpurun): bench lines, kernel stats, PMC passes (each in its own run).
# usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...; summaries are copied into profiles/ by hand afterwards
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r02}
O=gpurun_out/$T
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --mode train --precision bf16 > $O/bench_train_bf16.json 2>> $O/bench.err
python bench.py --mode train --precision fp32 > $O/bench_train_fp32.json 2>> $O/bench.err
python bench.py --mode train --precision bf16x6 > $O/bench_train_bf16x6.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o fwd -- python3 bench.py --no-cpu-baseline --no-fp32 --no-configs --settle 0 --steps 10 --warmup 2 > $O/ks_fwd.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kst -o train -- python3 bench.py --mode train --precision bf16 --no-graph --settle-steps 2 --steps 10 --warmup 0 > $O/ks_train.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 bench.py --no-cpu-baseline --no-fp32 --no-configs --settle 0 --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 bench.py --no-cpu-baseline --no-fp32 --no-configs --settle 0 --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1 -o p -- python3 bench.py --no-cpu-baseline --no-fp32 --no-configs --settle 0 --steps 3 --warmup 1 > $O/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o p -- python3 bench.py --no-cpu-baseline --no-fp32 --no-configs --settle 0 --steps 3 --warmup 1 > $O/pmc_sq2.log 2>&1
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/fwd_kernel_stats.csv
cp $(find $O/kst -name "*kernel_stats.csv" | head -1) $O/train_bf16_kernel_stats.csv     # 12 steps in all (2 + 10)
bash tools/step_trace.sh $T bf16 > /dev/null 2>&1; cp $O/step_bf16.txt $O/train_step_bf16_trace.txt
PMC_STEPS=4 python tools/pmc_summary.py $O/pmc.json 65536 256 $(find $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2 -name "*counter_collection.csv") > $O/pmc_summary.log 2>&1
cp profiles/traffic.json $O/traffic.json
python tools/kstats.py $O/fwd_kernel_stats.csv 14; tail -3 $O/pmc_summary.log; cut -c1-400 $O/bench.json
