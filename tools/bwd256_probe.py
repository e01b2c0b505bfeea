"""moda_bwd256_layer (bwd256_fused.hip; PROBE_W = 256 or 128) called directly against torch on the same bf16 operands: where do dX / dW / db differ?
usage: python tools/bwd256_probe.py [M ...]"""
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import torch

from moda_amd import _lib

lib = _lib.load()
import subprocess
sym = [l.split()[-1] for l in subprocess.check_output(["nm", "-D", _lib.LIB_PATH], text=True).splitlines() if "moda_bwd256_layer" in l][0]
f = getattr(lib, sym)     # internal C++ entry (moda_dev.h), not part of the C ABI
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p,
              ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
torch.manual_seed(0)
W = int(os.environ.get("PROBE_W", "256"))
for M in [int(v) for v in sys.argv[1:]] or [64, 100, 8192, 8320, 20000, 262144]:
    dz = torch.randn(M, W, device="cuda").bfloat16()
    x = torch.relu(torch.randn(M, W, device="cuda")).bfloat16()
    w = (torch.randn(W, W, device="cuda") / 16).bfloat16()
    guard = 4096
    dxbuf = torch.full((M * W + guard,), 7.0, device="cuda", dtype=torch.bfloat16)
    gW = torch.zeros(W, W, device="cuda")
    gb = torch.zeros(W, device="cuda")
    rc = f(W, dz.data_ptr(), W, x.data_ptr(), W, w.data_ptr(), W, dxbuf.data_ptr(), W, gW.data_ptr(), W, gb.data_ptr(), M, None)
    torch.cuda.synchronize()
    dx = dxbuf[:M * W].view(M, W).float()
    ref_dx = ((dz.float() @ w.float()) * (x > 0)).bfloat16().float()
    ref_gW = dz.float().t() @ x.float()
    ref_gb = dz.float().sum(0)
    bad = ((dx - ref_dx).abs() > 0.02 * ref_dx.abs() + 1e-2)
    rows = bad.any(1).nonzero().reshape(-1)
    print(f"M {M}: rc {rc}; dX bad entries {int(bad.sum())} in {rows.numel()} rows (first {rows[:6].tolist()}, tiles {sorted(set((rows // 64).tolist()))[:8]}); "
          f"bad by column half {[int(bad[:, :W // 2].sum()), int(bad[:, W // 2:].sum())]}; dW rel {float((gW - ref_gW).norm() / ref_gW.norm()):.2e}; "
          f"db rel {float((gb - ref_gb).norm() / ref_gb.norm()):.2e}; guard intact {bool((dxbuf[M * W:] == 7.0).all())}")
    if M >= 8192:
        ts = []
        for _ in range(30):
            s_ = torch.cuda.Event(enable_timing=True); e_ = torch.cuda.Event(enable_timing=True)
            s_.record()
            f(W, dz.data_ptr(), W, x.data_ptr(), W, w.data_ptr(), W, dxbuf.data_ptr(), W, gW.data_ptr(), W, gb.data_ptr(), M, None)
            e_.record(); torch.cuda.synchronize(); ts.append(s_.elapsed_time(e_) * 1e3)
        ts.sort()
        print(f"   launch: median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us  ({M * W * 2 * 3 / ts[len(ts) // 2] / 1e6:.2f} TB/s of dZ + X + dX)")
