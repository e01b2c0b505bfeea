"""moda_bwd256_layer (bwd256_fused.hip) called directly against torch on the same bf16 operands: where do dX / dW / db differ?
usage: python tools/bwd256_probe.py [M ...]"""
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import torch

from moda_amd import _lib

lib = _lib.load()
f = getattr(lib, "_Z17moda_bwd256_layerPKvxS0_xS0_xPvxPfxS2_xS1_")     # internal C++ entry (moda_dev.h), not part of the C ABI
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p,
              ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
torch.manual_seed(0)
for M in [int(v) for v in sys.argv[1:]] or [64, 100, 8192, 8320, 20000, 262144]:
    dz = torch.randn(M, 256, device="cuda").bfloat16()
    x = torch.relu(torch.randn(M, 256, device="cuda")).bfloat16()
    w = (torch.randn(256, 256, device="cuda") / 16).bfloat16()
    guard = 4096
    dxbuf = torch.full((M * 256 + guard,), 7.0, device="cuda", dtype=torch.bfloat16)
    gW = torch.zeros(256, 256, device="cuda")
    gb = torch.zeros(256, device="cuda")
    rc = f(dz.data_ptr(), 256, x.data_ptr(), 256, w.data_ptr(), 256, dxbuf.data_ptr(), 256, gW.data_ptr(), 256, gb.data_ptr(), M, None)
    torch.cuda.synchronize()
    dx = dxbuf[:M * 256].view(M, 256).float()
    ref_dx = ((dz.float() @ w.float()) * (x > 0)).bfloat16().float()
    ref_gW = dz.float().t() @ x.float()
    ref_gb = dz.float().sum(0)
    bad = ((dx - ref_dx).abs() > 0.02 * ref_dx.abs() + 1e-2)
    rows = bad.any(1).nonzero().reshape(-1)
    print(f"M {M}: rc {rc}; dX bad entries {int(bad.sum())} in {rows.numel()} rows (first {rows[:6].tolist()}, tiles {sorted(set((rows // 64).tolist()))[:8]}); "
          f"bad by column half {[int(bad[:, :128].sum()), int(bad[:, 128:].sum())]}; dW rel {float((gW - ref_gW).norm() / ref_gW.norm()):.2e}; "
          f"db rel {float((gb - ref_gb).norm() / ref_gb.norm()):.2e}; guard intact {bool((dxbuf[M * 256:] == 7.0).all())}")
    if M >= 8192:
        ts = []
        for _ in range(30):
            s_ = torch.cuda.Event(enable_timing=True); e_ = torch.cuda.Event(enable_timing=True)
            s_.record()
            f(dz.data_ptr(), 256, x.data_ptr(), 256, w.data_ptr(), 256, dxbuf.data_ptr(), 256, gW.data_ptr(), 256, gb.data_ptr(), M, None)
            e_.record(); torch.cuda.synchronize(); ts.append(s_.elapsed_time(e_) * 1e3)
        ts.sort()
        print(f"   launch: median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us  ({M * 256 * 2 * 3 / ts[len(ts) // 2] / 1e6:.2f} TB/s of dZ + X + dX)")
