"""Micro-benchmark of the training GEMMs (old strided kernel vs the pipelined one) at the MLP's shapes."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
if len(sys.argv) > 2:
    os.environ["MODA_HIPCC_FLAGS"] = sys.argv[2]
    from moda_amd import build
    build.build(force=True, verbose=False)
from moda_amd import _lib as L, autograd as A
torch.manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def old(a, b, out, acc=False, split_k=1):
    Mm, K = a.shape; N = b.shape[1]
    L.call("moda_gemm_f32", L.ptr(a), a.stride(0), a.stride(1), L.ptr(b), b.stride(0), b.stride(1), L.ptr(out), out.stride(0),
           Mm, N, K, None, 0, None, int(acc), split_k, L.stream())

print("variant:", os.environ.get("MODA_HIPCC_FLAGS", "default"))
for (K, N) in ((256, 256), (64, 256), (320, 256), (256, 128), (128, 128)):
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; dz = torch.randn(M, N, device="cuda")
    y = torch.empty(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda"); dW = torch.zeros(N, K, device="cuda")
    fl = 2.0 * M * K * N / 1e9
    sk = max(1, min(M // 128, 2048 // (((N + 127) // 128) * ((K + 127) // 128))))
    res = []
    for name, fo, fn in (("fwd", lambda: old(x, W.t(), y), lambda: A.gemm(x, W.t(), out=y)),
                         ("dX ", lambda: old(dz, W, dx), lambda: A.gemm(dz, W, out=dx)),
                         ("dW ", lambda: old(dz.t(), x, dW, True, sk), lambda: A.gemm(dz.t(), x, out=dW, accumulate=True, split_k=sk))):
        to, tn = timeit(fo), timeit(fn)
        res.append(f"{name} old {to*1e3:7.0f}us {fl/to:6.1f}TF | new {tn*1e3:7.0f}us {fl/tn:6.1f}TF")
    ref = x @ W.t()
    A.gemm(x, W.t(), out=y)
    err = float((y - ref).abs().max() / ref.abs().max())
    print(f"M={M} K={K} N={N}: " + " ; ".join(res) + f" ; fwd err {err:.1e}")
