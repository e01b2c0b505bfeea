"""Diagnostic: per-phase s_memtime shares of the fused coarse MLP kernel (build with -DMODA_STAMPS)."""
import sys, os, ctypes
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ["MODA_HIPCC_FLAGS"] = "-DMODA_STAMPS " + (sys.argv[1] if len(sys.argv) > 1 else "")
from moda_amd import build
build.build(force=True, verbose=False)
import numpy as np, torch
import moda_amd
from moda_amd import synth, _lib
from gpu_helpers import T, nerf_from_params
torch.set_grad_enabled(False)
N, S = 65536, 256
NET = os.environ.get("MODA_STAMP_NET", "coarse")
xyz = torch.from_numpy(np.float32(0.3) * synth.normal(5, "mb/xyz", (4096 * 16, 3))).cuda().repeat(N * S // (4096 * 16), 1).contiguous()
if NET == "coarse":
    kw = dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False)
    p = synth.nerf_params(5, "mb/coarse", D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3)
    m = nerf_from_params(p, **kw)
    dirs = T(synth.normal(5, "mb/dir", (N, 91)))
    run = lambda: m.fused(xyz, dir_src=dirs, precision="bf16")
else:
    kw = dict(D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=25, raw_feat=True)
    p = synth.nerf_params(5, "mb/skin", D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=25)
    m = nerf_from_params(p, **kw)
    code = T(synth.normal(5, "mb/code", (N, 128)))
    run = lambda: m.fused(xyz.view(N, S, 3), code=code, precision="bf16", out_tr_S=S)
for _ in range(2):
    run()
torch.cuda.synchronize()
_s = torch.cuda.Event(enable_timing=True); _e = torch.cuda.Event(enable_timing=True)
_s.record(); run(); _e.record(); torch.cuda.synchronize()
print(f"[{NET}] one launch (stamped build): {_s.elapsed_time(_e):.3f} ms")
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
lib.moda_dbg_read_stamps.restype = ctypes.c_int
assert lib.moda_dbg_read_stamps(buf) == 0
v = np.array(list(buf), dtype=np.float64)
names = ["0 load+PE+rowbias", "1", "2 layer 1", "3", "4 layers 2..D", "5", "6", "7 sigma+final", "8 dir", "9 rgb", "10 store", "11", "12", "13", "14", "15 loop"]
main = v[:11].sum() + v[15]
ntiles_per_wg = N * S / 256 / 256
print(f"total stamped cycles per WG-tile (wave 0): {main / 256 / ntiles_per_wg:.0f}")
for n, x in zip(names, v):
    if x: print(f"  {n:26s} {x / 256 / ntiles_per_wg:9.0f} cyc/tile  {100 * x / main:5.1f}%")
