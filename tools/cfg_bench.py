"""Timing of the other BASELINE configurations' forward renders (not the headline metric):
cfg3 = 36 bones + symmetric-shape branch, cfg5 = hierarchical (128 + 128) sampling + CSE feature head.
usage: python tools/cfg_bench.py [cfg3|cfg5|cfg2] [rays] [precision]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import moda_amd
from moda_amd import synth
from gpu_helpers import make_models, make_opts, rays_to_gpu

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
S = 256
B = 36 if cfg == "cfg3" else 25
models, emb = make_models(0, B, with_feat=(cfg == "cfg5"), perturb_bones=(cfg == "cfg3"))
rays = rays_to_gpu(synth.make_rays(1000, N, B, rays_per_frame=256))
opts = make_opts(symm_shape=(cfg == "cfg3"))
kw = dict(N_samples=S, perturb=0, noise_std=0.0, opts=opts, img_size=512, use_fine=(cfg == "cfg5"))
moda_amd.set_precision(sys.argv[3] if len(sys.argv) > 3 else "bf16")
with torch.no_grad():
    for _ in range(3):
        moda_amd.render_rays(models, emb, rays, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        r = moda_amd.render_rays(models, emb, rays, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print(f"{cfg}: {N} rays x {S} samples, {B} bones: {dt*1e3:.2f} ms per call = {N/dt/1e6:.2f} M rays/s; img mean {float(r['img_coarse'].mean()):.5f}")
