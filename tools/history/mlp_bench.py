"""Micro-benchmark (timing only; parity lives in tests/) of the fused PE+MLP kernel (coarse 8x256 and skin 5x64) for a given build variant.
usage: python tools/mlp_bench.py ["<hipcc flags>"] [--rays N]"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
variant = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else ""
os.environ["MODA_HIPCC_FLAGS"] = variant
from moda_amd import build
build.build(force=True, verbose=False)
import moda_amd
from moda_amd import synth
from gpu_helpers import T, nerf_from_params
torch.set_grad_enabled(False)
N, S = 65536, 256
M = N * S
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts)), float(np.min(ts))
xyz = torch.from_numpy(np.float32(0.3) * synth.normal(5, "mb/xyz", (4096 * 16, 3))).cuda().repeat(M // (4096 * 16), 1).contiguous()
out = []
for name, kw, macs, code_c, dir_c in (
        ("coarse", dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False), 601600, 0, 91),
        ("skin", dict(D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=25, raw_feat=True), 47840, 128, 0)):
    pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
    p = synth.nerf_params(5, "mb/" + name, **pk)
    m = nerf_from_params(p, **kw)
    code = T(synth.normal(5, "mb/code", (N, code_c))) if code_c else None
    dirs = T(synth.normal(5, "mb/dir", (N, dir_c))) if dir_c else None
    # the skin net as render_rays launches it: channel-major (N, B, S) logits for the warp kernel
    tr = dict(out_tr_S=S) if name == "skin" else {}
    med, mn = timeit(lambda: m.fused(xyz.view(N, S, 3), code=code, dir_src=dirs, precision="bf16", **tr))
    out.append(f"{name}: {med:7.3f} ms (min {mn:7.3f}) = {2*macs*M/med/1e9:7.1f} TFLOP/s algorithmic")
# the exact-fp32 parity kernels (what every 1e-4 parity test runs): a regression here does not show in the bf16 lines
m = nerf_from_params(synth.nerf_params(5, "mb/coarse", D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3),
                     D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False)
n32 = 8192
d32 = T(synth.normal(5, "mb/dir", (N, 91)))[:n32].contiguous()
med, mn = timeit(lambda: m.fused(xyz[:n32 * S].view(n32, S, 3), dir_src=d32, precision="fp32"), n=3)
out.append(f"coarse fp32 ({n32} rays): {med:7.3f} ms = {2*601600*n32*S/med/1e9:6.1f} TFLOP/s")
print(f"[{variant or 'default'}] " + " | ".join(out))
