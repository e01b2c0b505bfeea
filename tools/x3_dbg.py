"""debug: where does the split-bf16 dX / forward / dW form differ from float64?  python tools/x3_dbg.py"""
import numpy as np, torch
import moda_amd
from moda_amd import autograd as A
torch.manual_seed(0)
for (M, K, N) in [(16500, 64, 64), (300, 64, 256), (300, 64, 64), (16500, 256, 64), (1000, 128, 64), (129, 64, 64), (4099, 64, 64)]:
    dz = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.1
    moda_amd.set_train_precision("bf16x3")
    dx = A.gemm(dz, w)
    moda_amd.set_train_precision("fp32")
    ref = dz.double() @ w.double()
    e = (dx.double() - ref).abs()
    bad = (e > 1e-4 * ref.abs().max()).nonzero()
    print((M, K, N), "max err", float(e.max() / ref.abs().max()), "bad", bad.shape[0],
          "rows", sorted(set((bad[:, 0] // 128).tolist()))[:10], "cols", sorted(set(bad[:, 1].tolist()))[:16],
          "first", bad[:5].tolist())
