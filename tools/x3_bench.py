"""Timing of the precision modes on cfg2-shaped rays: python tools/x3_bench.py [rays] [modes...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import moda_amd
from moda_amd import synth, _lib
from gpu_helpers import make_models, make_opts, rays_to_gpu
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
modes = sys.argv[2:] or ["bf16x3", "fp32", "bf16"]
S, B = 256, 25
models, emb = make_models(0, B)
rays = rays_to_gpu(synth.make_rays(1000, N, B, rays_per_frame=256))
kw = dict(N_samples=S, perturb=0, noise_std=0.0, opts=make_opts(), img_size=512)
ref = None
for mode in modes:
    moda_amd.set_precision(mode)
    with torch.no_grad():
        for _ in range(2):
            r = moda_amd.render_rays(models, emb, rays, **kw)
        torch.cuda.synchronize()
        _lib.PROFILE = {}
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            r = moda_amd.render_rays(models, emb, rays, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        prof, _lib.PROFILE = _lib.PROFILE, None
    ks = {t: float(np.mean([s.elapsed_time(e) for s, e, _ in ev])) for t, ev in prof.items()}
    print(f"{mode}: {N} rays x {S}: {dt*1e3:.2f} ms = {N/dt/1e6:.3f} M rays/s; kernels ms {ks}")
    if mode == "fp32":
        ref = {k: v.clone() for k, v in r.items() if torch.is_tensor(v)}
    elif ref is not None:
        print("   vs fp32:", {k: f"{float((r[k]-ref[k]).abs().max()/ref[k].abs().max()):.1e}" for k in ("img_coarse", "depth_rnd", "sil_coarse", "xyz_canonical_vis", "frame_cyc_dis")})
