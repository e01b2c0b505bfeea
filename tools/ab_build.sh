#!/bin/bash
# usage (build container): tools/ab_build.sh <name> "<extra hipcc flags>" [source.hip ...]   (default source: mlp_fused.hip)
# Compiles the given sources with the extra flags and links them with the DEFAULT objects of the other sources into
# moda_amd/lib/ab/<name>.so -- an A/B variant library that travels to the GPU box and is loaded with MODA_LIB_PATH.
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2
srcs=${@:-mlp_fused.hip}
mkdir -p moda_amd/lib/ab
objs=""
for s in mlp_fused render_kernels train_kernels gemm_bf16 gemm_x3 bwd64_chain bwd256_fused loss_kernels prep_kernels; do
  if echo " $srcs " | grep -q " $s.hip "; then
    extra=""
    [ "$s" = "mlp_fused" ] && extra="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -c moda_amd/csrc/$s.hip -o moda_amd/lib/ab/${name}_$s.o $extra $flags
    objs="$objs moda_amd/lib/ab/${name}_$s.o"
  else
    objs="$objs moda_amd/lib/$s.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o moda_amd/lib/ab/$name.so $objs
echo "built moda_amd/lib/ab/$name.so"
