# split-K target sweep of the bf16-storage dW GEMM (MODA_GEMM3_BLOCKS) through tools/gemm_bench.py; usage (GPU box): bash tools/gemm_sweep.sh
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for b in 128 256 512 1024; do echo "MODA_GEMM3_BLOCKS=$b"; MODA_GEMM3_BLOCKS=$b python tools/gemm_bench.py; done
