"""Diagnostic: a batch rendered in one call vs in chunks must agree bit for bit (rays are independent)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import moda_amd
from moda_amd import synth, rendering as R
from gpu_helpers import make_models, make_opts, rays_to_gpu

N, S, B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 256, 25
models, emb = make_models(0, B)
rays = rays_to_gpu(synth.make_rays(1000, N, B, rays_per_frame=256))
keys = ("xyz_canonical_vis", "frame_cyc_dis", "img_coarse", "depth_rnd", "sil_coarse")
for prec in ("bf16", "fp32"):
    for fused in (True, False):
        if prec == "fp32" and fused:
            continue
        R.FUSED_WARP = fused
        moda_amd.set_precision(prec)
        n = N if prec == "bf16" else min(N, 16384)
        rr = {k: v[:n] for k, v in rays.items()}
        with torch.no_grad():
            full = moda_amd.render_rays(models, emb, rr, N_samples=S, perturb=0, noise_std=0.0, opts=make_opts(), img_size=512)
            full2 = moda_amd.render_rays(models, emb, rr, N_samples=S, perturb=0, noise_std=0.0, opts=make_opts(), img_size=512)
            c = n // 8
            sub = {k: v[c:2 * c] for k, v in rr.items()}
            part = moda_amd.render_rays(models, emb, sub, N_samples=S, perturb=0, noise_std=0.0, opts=make_opts(), img_size=512)
        for k in keys:
            a, b = full[k][c:2 * c], part[k]
            d = (a - b).abs()
            rep = (full[k] - full2[k]).abs().max().item()
            print(f"{prec} fused={fused} {k}: chunk-vs-full max abs diff {d.max().item():.3e} ({int((d > 0).sum())} of {d.numel()} differ); "
                  f"repeat-call diff {rep:.1e}")
