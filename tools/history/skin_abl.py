"""Timing of the 5x64 skin MLP launch (channel-major output, per-ray code) for a given build variant."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
os.environ["MODA_HIPCC_FLAGS"] = sys.argv[1] if len(sys.argv) > 1 else ""
from moda_amd import build
build.build(force=True, verbose=False)
from moda_amd import synth
from gpu_helpers import T, nerf_from_params
torch.set_grad_enabled(False)
N, S = 65536, 256
kw = dict(D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=25, raw_feat=True)
p = synth.nerf_params(5, "mb/skin", D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=25)
m = nerf_from_params(p, **kw)
xyz = (0.3 * torch.randn(N, S, 3, device="cuda")).contiguous()
code = T(synth.normal(5, "mb/code", (N, 128)))
run = lambda: m.fused(xyz, code=code, precision="bf16", out_tr_S=S)
run(); torch.cuda.synchronize()
ts = []
for _ in range(7):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); run(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
print(f"[{os.environ['MODA_HIPCC_FLAGS'] or 'default'}] skin MLP (incl. 2 code folds): {np.median(ts):.3f} ms (min {min(ts):.3f})")
