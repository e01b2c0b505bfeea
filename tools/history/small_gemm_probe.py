"""probe: the per-ray code-gradient product of the training backward (K = rays, 64 x 128 output) through moda_gemm_f32_ex"""
import torch
from moda_amd import _lib as L
import importlib.util, sys, os
spec = importlib.util.spec_from_file_location("g32", os.path.join(os.path.dirname(__file__), "gemm32_bench.py"))
DEV = "cuda"
def run(A_, sam, sak, B_, sbk, sbn, C_, Mm, Nn, Kk, flags, acc=0, split=1):
    d = L.GemmDesc(A=A_.data_ptr(), sam=sam, sak=sak, A2=None, sam2=0, K1=Kk, B=B_.data_ptr(), sbk=sbk, sbn=sbn,
                   C=C_.data_ptr(), ldc=C_.stride(0), M=Mm, N=Nn, K=Kk, bias=None, rowbias=None, ld_rowbias=0, rows_per_bias=1,
                   mask_src=None, ld_mask=0, act=0, accumulate=acc, split_k=split, reserved=flags, a_sum=None, mask_bits=None, ld_bits=0)
    L.call("moda_gemm_f32_ex", L._c.byref(d), L.stream())
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
R, W, C = 2048, 64, 128
drb = torch.randn(R, W, device=DEV); code = torch.randn(R, C, device=DEV); out = torch.zeros(W, 191, device=DEV)
wgt = torch.randn(W, 191, device=DEV); dcode = torch.zeros(R, C, device=DEV)
for fl in (0, 1, 64):
    for sp in (1, 8, 64):
        t = timeit(lambda: run(drb, 1, W, code, C, 1, out[:, 63:], W, C, R, fl, acc=1, split=sp))
        print(f"flags {fl} dWcode split {sp}: {t:.1f} us")
    t = timeit(lambda: run(drb, W, 1, wgt[:, 63:], 191, 1, dcode, R, C, W, fl, acc=2))
    print(f"flags {fl} d_code: {t:.1f} us")
