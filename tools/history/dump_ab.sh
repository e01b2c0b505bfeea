# A/B of build flags on the training step: usage bash tools/dump_ab.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for f in "$@"; do
  echo "== flags: $f"
  MODA_HIPCC_FLAGS="$f" python -m moda_amd.build --force > /dev/null 2>&1
  python bench.py --mode train --precision bf16 --steps 60 | cut -c150-260
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dab -o t -- python3 bench.py --mode train --precision bf16 --no-graph --settle 0 --steps 6 --warmup 2 > /dev/null 2>&1
  python tools/kstats.py $(find gpurun_out/dab -name "*kernel_stats.csv" | head -1) 40 | grep "mlp_fused"
  rm -rf gpurun_out/dab
done
