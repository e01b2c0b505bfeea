"""Times the GEMM forms of the bf16-storage training backward (gemm_bf16.hip / gemm2) on one GPU.
usage: python tools/gemm_bench.py [M=262144]     (env MODA_GEMM3=0 routes everything to the generic kernel)"""
import sys

import torch

from moda_amd import _lib as L

DEV = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
BF, FA, FB, FC, FM = 1, 2, 4, 8, 16


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
    dz = torch.randn(M, W, device=DEV).bfloat16()
    h = torch.randn(M, W, device=DEV).bfloat16()
    w = torch.randn(W, W, device=DEV)
    out = torch.empty(M, W, device=DEV, dtype=torch.bfloat16)
    dW = torch.zeros(W, W, device=DEV)
    db = torch.zeros(W, device=DEV)
    pe = torch.randn(M, 64, device=DEV)
    dpe = torch.empty(M, 64, device=DEV)
    t = timeit(lambda: run(dz, W, 1, w, W, 1, out, M, W, W, BF | FA | FC | FM, mask_=h))
    gb = M * W * 6 / 1e9
    print(f"W={W} dX  (bf16 dZ, mask, bf16 out)     {t:8.1f} us  {gb / t * 1e3:6.2f} TB/s")
    wb = w.bfloat16()
    t = timeit(lambda: run(dz, W, 1, wb, W, 1, out, M, W, W, BF | FA | FB | FC | FM, mask_=h))
    print(f"W={W} dX  (same, bf16 weights)          {t:8.1f} us  {gb / t * 1e3:6.2f} TB/s")
    t = timeit(lambda: run(dz, 1, W, h, W, 1, dW, W, W, M, BF | FA | FB, acc=1, split=1, asum=db))
    gb = M * W * 4 / 1e9
    print(f"W={W} dW  (bf16 dZ^T bf16 X + db)       {t:8.1f} us  {gb / t * 1e3:6.2f} TB/s")
    t = timeit(lambda: run(dz, 1, W, pe, 64, 1, dW, W, 63, M, BF | FA, acc=1, split=1))
    gb = M * (W * 2 + 256) / 1e9
    print(f"W={W} dW  (bf16 dZ^T fp32 pe)           {t:8.1f} us  {gb / t * 1e3:6.2f} TB/s")
    t = timeit(lambda: run(dz, W, 1, w[:, :64].contiguous(), 64, 1, dpe, M, 64, W, BF | FA))
    gb = M * (W * 2 + 256) / 1e9
    print(f"W={W} dpe (bf16 dZ @ W -> fp32 64)      {t:8.1f} us  {gb / t * 1e3:6.2f} TB/s")
