# Round profile of bench.py on one MI355X (run through gpurun): bench lines, kernel stats, PMC passes (each in its own run).
# usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...; summaries are copied into profiles/ by hand afterwards
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r06}
O=gpurun_out/$T
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --precision bf16 --no-configs > $O/bench_bf16.json 2>> $O/bench.err
for P in bf16 bf16x6 fp32; do python bench.py --mode train --precision $P > $O/bench_train_$P.json 2>> $O/bench.err; done
python bench.py --mode train --precision bf16 --rays 8192 --samples 256 --steps 30 --settle-steps 12 > $O/bench_train_bf16_cfg4_8192x256.json 2>> $O/bench.err
python bench.py --mode train --precision bf16 --fine --unc --steps 30 --settle-steps 12 > $O/bench_train_bf16_cfg5_2048x128.json 2>> $O/bench.err
python bench.py --mode train --precision bf16 --fine --unc --rays 8192 --samples 256 --steps 30 --settle-steps 12 > $O/bench_train_bf16_cfg5_8192x256.json 2>> $O/bench.err
FW="--no-cpu-baseline --no-fp32 --no-configs --settle 0"
for P in fp16 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$P -o fwd -- python3 bench.py --precision $P $FW --steps 10 --warmup 2 > $O/ks_fwd_$P.log 2>&1
  cp $(find $O/ks_$P -name "*kernel_stats.csv" | head -1) $O/fwd_${P}_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$P -o p -- python3 bench.py --precision $P $FW --steps 3 --warmup 1 > $O/pmc_fetch_$P.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$P -o p -- python3 bench.py --precision $P $FW --steps 3 --warmup 1 > $O/pmc_write_$P.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1_$P -o p -- python3 bench.py --precision $P $FW --steps 3 --warmup 1 > $O/pmc_sq1_$P.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2_$P -o p -- python3 bench.py --precision $P $FW --steps 3 --warmup 1 > $O/pmc_sq2_$P.log 2>&1
  PMC_MODE=$P PMC_STEPS=4 python tools/pmc_summary.py $O/pmc_$P.json 65536 256 $(find $O/pmc_fetch_$P $O/pmc_write_$P $O/pmc_sq1_$P $O/pmc_sq2_$P -name "*counter_collection.csv") > $O/pmc_summary_$P.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kst -o train -- python3 bench.py --mode train --precision bf16 --no-graph --settle-steps 2 --steps 10 --warmup 0 > $O/ks_train.log 2>&1
cp $(find $O/kst -name "*kernel_stats.csv" | head -1) $O/train_bf16_kernel_stats.csv     # 12 steps in all (2 + 10)
bash tools/step_trace.sh $T bf16 cfg4_2048x128 > /dev/null 2>&1; cp $O/step_cfg4_2048x128.txt $O/train_step_bf16_trace.txt
bash tools/step_trace.sh $T bf16 cfg4_8192x256 --rays 8192 --samples 256 > /dev/null 2>&1; cp $O/step_cfg4_8192x256.txt $O/train_8192x256_cfg4_bf16_trace.txt
bash tools/step_trace.sh $T bf16 cfg5_2048x128 --fine --unc > /dev/null 2>&1; cp $O/step_cfg5_2048x128.txt $O/train_2048x128_cfg5_bf16_trace.txt
bash tools/step_trace.sh $T bf16 cfg5_8192x256 --fine --unc --rays 8192 --samples 256 > /dev/null 2>&1; cp $O/step_cfg5_8192x256.txt $O/train_8192x256_cfg5_bf16_trace.txt
bash tools/pmc_train_step.sh $T bf16 > $O/pmc_train_a.log 2>&1
bash tools/pmc_train_step.sh $T bf16x6 > $O/pmc_train_b.log 2>&1
bash tools/pmc_train_step.sh $T bf16 _cfg4_8192x256 "cfg4 training step (8192 rays x 256 samples)" --rays 8192 --samples 256 > $O/pmc_train_c.log 2>&1
bash tools/pmc_train_step.sh $T bf16x6 _cfg4_8192x256 "cfg4 training step (8192 rays x 256 samples)" --rays 8192 --samples 256 > $O/pmc_train_d.log 2>&1
bash tools/pmc_train_step.sh $T bf16 _cfg5_2048x128 "cfg5 training step (2048 rays x 64+64 samples, use_fine + unc)" --fine --unc > $O/pmc_train_e.log 2>&1
bash tools/pmc_train_step.sh $T bf16x6 _cfg5_2048x128 "cfg5 training step (2048 rays x 64+64 samples, use_fine + unc)" --fine --unc > $O/pmc_train_e2.log 2>&1
bash tools/pmc_train_step.sh $T bf16 _cfg5_8192x256 "cfg5 training step (8192 rays x 128+128 samples, use_fine + unc)" --fine --unc --rays 8192 --samples 256 > $O/pmc_train_f.log 2>&1
bash tools/pmc_train_step.sh $T bf16x6 _cfg5_8192x256 "cfg5 training step (8192 rays x 128+128 samples, use_fine + unc)" --fine --unc --rays 8192 --samples 256 > $O/pmc_train_g.log 2>&1
cp profiles/traffic.json $O/traffic.json
find $O -name "*.csv" -size +3M -delete            # raw traces stay on the box; the summaries travel
python tools/kstats.py $O/fwd_fp16_kernel_stats.csv 14; tail -3 $O/pmc_summary_fp16.log; cut -c1-400 $O/bench.json; cat $O/pmc_train_*.log | grep "train step"
