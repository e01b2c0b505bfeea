"""Times the three GEMM forms of the split-bf16 training modes (gemm_x3.hip) at the cfg4 layer shape.
usage: python tools/gemm_x3_bench.py [mode=bf16x6] [M=262144] [W=256] [reps=30]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import moda_amd
from moda_amd import autograd as A

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
W = int(sys.argv[3]) if len(sys.argv) > 3 else 256
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
DEV = "cuda"
x = torch.randn(M, W, device=DEV)
w = torch.randn(W, W, device=DEV) * 0.05
b = torch.randn(W, device=DEV)
dz = torch.randn(M, W, device=DEV)
out = torch.empty(M, W, device=DEV)
dW = torch.zeros(W, W, device=DEV)
moda_amd.set_train_precision(mode)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


nm = 6 if mode == "bf16x6" else 3
fl = 2.0 * M * W * W * nm
for tag, fn, gb in (("forward (bias, ReLU)", lambda: A.gemm(x, w.t(), bias=b, act=1, out=out), M * W * 8 / 1e9),
                    ("dX (fp32 mask)", lambda: A.gemm(dz, w, mask_src=x, out=out), M * W * 12 / 1e9),
                    ("dX (no mask)", lambda: A.gemm(dz, w, out=out), M * W * 8 / 1e9),
                    ("dW (atomics)", lambda: A.gemm(dz.t(), x, out=dW, accumulate=True, split_k=8), M * W * 8 / 1e9)):
    t = timeit(fn)
    print(f"{mode} W={W} M={M} {tag:22s} {t:8.1f} us  {gb / t * 1e3:5.2f} TB/s  {fl / t / 1e9:6.3f} PFLOP/s executed")
