# round-6 baseline: the training step at the shapes SURVEY 8(d) names, before any optimisation
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --mode train --steps 30 --warmup 5 --settle-steps 12 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; cut -c1-300 $O/$name.json; tail -3 $O/$name.err; }
run cfg4_2048x128_bf16 --precision bf16
run cfg4_8192x256_bf16 --precision bf16 --rays 8192 --samples 256
run cfg4_8192x256_bf16x6 --precision bf16x6 --rays 8192 --samples 256
run cfg5_2048x128_bf16 --precision bf16 --fine --unc
run cfg5_8192x256_bf16 --precision bf16 --fine --unc --rays 8192 --samples 256
run cfg5_8192x256_bf16x6 --precision bf16x6 --fine --unc --rays 8192 --samples 256
bash tools/step_trace.sh r06a bf16 cfg4_8192x256 --rays 8192 --samples 256 > /dev/null 2>&1
bash tools/step_trace.sh r06a bf16 cfg5_8192x256 --rays 8192 --samples 256 --fine --unc > /dev/null 2>&1
bash tools/step_trace.sh r06a bf16 cfg4_2048x128 > /dev/null 2>&1
grep "^#" $O/step_cfg4_8192x256.txt | head -40
