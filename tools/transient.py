"""Diagnostic: per-launch time of the fused warp / coarse kernels over the first steps of a process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import moda_amd
from moda_amd import synth, _lib
from gpu_helpers import make_models, make_opts, rays_to_gpu
N, S, B = 65536, 256, 25
models, emb = make_models(0, B)
rays = rays_to_gpu(synth.make_rays(1000, N, B, rays_per_frame=256))
moda_amd.set_precision("bf16")
_lib.PROFILE = {}
with torch.no_grad():
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
        moda_amd.render_rays(models, emb, rays, N_samples=S, perturb=0, noise_std=0.0, opts=make_opts(), img_size=512)
torch.cuda.synchronize()
for tag, ev in _lib.PROFILE.items():
    t = np.array([s.elapsed_time(e) for s, e, _ in ev])
    if "warp" in tag:
        t = t.reshape(-1, 2).sum(1)
    print(tag, "per step, every 10th:", np.round(t[::10], 2).tolist())
