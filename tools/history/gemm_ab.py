"""A/B of the generic GEMM's dW / dX forms (fp32 operands) between two builds of the library.
usage: python tools/gemm_ab.py <lib.so> [M]"""
import ctypes as C
import sys

import torch

import moda_amd.build as B
B.LIB_PATH = sys.argv[1]
import moda_amd._lib as L
L.LIB_PATH = sys.argv[1]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
DEV = "cuda"


def run(A_, sam, sak, B_, sbk, sbn, C_, Mm, Nn, Kk, flags, mask_=None, acc=0, split=1, asum=None):
    d = L.GemmDesc(A=A_.data_ptr(), sam=sam, sak=sak, A2=None, sam2=0, K1=Kk, B=B_.data_ptr(), sbk=sbk, sbn=sbn,
                   C=C_.data_ptr(), ldc=C_.stride(0), M=Mm, N=Nn, K=Kk, bias=None, rowbias=None, ld_rowbias=0,
                   rows_per_bias=1, mask_src=None if mask_ is None else mask_.data_ptr(),
                   ld_mask=0 if mask_ is None else mask_.stride(0), act=0, accumulate=acc, split_k=split, reserved=flags,
                   a_sum=None if asum is None else asum.data_ptr())
    L.call("moda_gemm_f32_ex", L._c.byref(d), L.stream())


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for W in (256, 64):
    dz = torch.randn(M, W, device=DEV)
    h = torch.randn(M, W, device=DEV)
    w = torch.randn(W, W, device=DEV)
    out = torch.empty(M, W, device=DEV)
    dW = torch.zeros(W, W, device=DEV)
    db = torch.zeros(W, device=DEV)
    for fl, nm in ((0, "fp32"), (1, "bf16 operands")):
        for sp in (128, 512) if W == 256 else (512,):
            t = timeit(lambda: run(dz, 1, W, h, W, 1, dW, W, W, M, fl, acc=1, split=sp, asum=db))
            print(f"W={W} {nm:14s} dW split {sp:4d}  {t:8.1f} us  {M * W * 8 / t / 1e6:6.2f} TB/s")
        t = timeit(lambda: run(dz, W, 1, w, W, 1, out, M, W, W, fl, mask_=h))
        print(f"W={W} {nm:14s} dX             {t:8.1f} us  {M * W * 12 / t / 1e6:6.2f} TB/s")
