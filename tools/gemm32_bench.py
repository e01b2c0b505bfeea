"""Times the three GEMM forms of a Linear layer of the training route with fp32 storage (forward, dX, dW) in each operand
mode of moda_gemm_f32_ex: exact fp32 (flags 0), split-bf16 (MODA_GEMM_BF16X6 = 64, MODA_GEMM_BF16X3 = 32), bf16 operands (MODA_GEMM_BF16 = 1).
usage: python tools/gemm32_bench.py [M=262144]"""
import sys

import torch

from moda_amd import _lib as L

DEV = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144


def run(A_, sam, sak, B_, sbk, sbn, C_, Mm, Nn, Kk, flags, mask_=None, acc=0, split=1, asum=None, bias=None, act=0):
    d = L.GemmDesc(A=A_.data_ptr(), sam=sam, sak=sak, A2=None, sam2=0, K1=Kk, B=B_.data_ptr(), sbk=sbk, sbn=sbn,
                   C=C_.data_ptr(), ldc=C_.stride(0), M=Mm, N=Nn, K=Kk, bias=None if bias is None else bias.data_ptr(),
                   rowbias=None, ld_rowbias=0, rows_per_bias=1, mask_src=None if mask_ is None else mask_.data_ptr(),
                   ld_mask=0 if mask_ is None else mask_.stride(0), act=act, accumulate=acc, split_k=split, reserved=flags,
                   a_sum=None if asum is None else asum.data_ptr(), mask_bits=None, ld_bits=0)
    L.call("moda_gemm_f32_ex", L._c.byref(d), L.stream())


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def split_k(Mm, rows, cols):
    tiles = ((rows + 127) // 128) * ((cols + 127) // 128 if cols > 64 else 1)
    return max(1, min(512 // tiles, Mm // 256))


for W in (256, 128, 64):
    x = torch.randn(M, W, device=DEV)
    dz = torch.randn(M, W, device=DEV)
    w = torch.randn(W, W, device=DEV) * 0.05
    b = torch.randn(W, device=DEV)
    out = torch.empty(M, W, device=DEV)
    dW = torch.zeros(W, W, device=DEV)
    db = torch.zeros(W, device=DEV)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    for name, fl in (("fp32", 0), ("bf16x6", 64), ("bf16x3", 32), ("bf16", 1)):
        t = timeit(lambda: run(x, W, 1, w, 1, W, out, M, W, W, fl, bias=b, act=1))
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        gb = M * W * 8 / 1e9
        line = f"W={W:3d} {name:6s} fwd {t:7.1f} us {gb / t * 1e3:5.2f} TB/s (err {err:.1e})"
        t = timeit(lambda: run(dz, W, 1, w, W, 1, out, M, W, W, fl, mask_=x))
        gb = M * W * 12 / 1e9
        line += f" | dX {t:7.1f} us {gb / t * 1e3:5.2f} TB/s"
        sp = split_k(M, W, W)
        t = timeit(lambda: run(dz, 1, W, x, W, 1, dW, W, W, M, fl, acc=1, split=sp, asum=db))
        gb = M * W * 8 / 1e9
        line += f" | dW(split {sp}) {t:7.1f} us {gb / t * 1e3:5.2f} TB/s"
        print(line, flush=True)
