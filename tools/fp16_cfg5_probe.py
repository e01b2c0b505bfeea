"""cfg5 (hierarchical 128 + 128 samples + the CSE feature network) in the fp16 mode: which precision may the two pieces outside the hot
loop run in?  For every combination of rendering.FP16_FEAT_PRECISION x rendering.FP16_PREPASS_PRECISION: the distance of every
output from the exact-fp32 mode (max |a - b| / max |b| and the per-element figure of tests/helpers.elem_err) on 8 192 rays, and the
time of a 65 536-ray call.   usage: python tools/fp16_cfg5_probe.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import numpy as np, torch
import moda_amd
from moda_amd import synth, rendering as R
from moda_amd.bench_support import make_models, make_opts, rays_to_gpu

torch.set_grad_enabled(False)
B, S = 25, 256
models, emb = make_models(0, B, with_feat=True)
rays_big = rays_to_gpu(synth.make_rays(1000, 65536, B, rays_per_frame=256))
rays = {k: v[:8192] for k, v in rays_big.items()}
kw = dict(N_samples=S, perturb=0, noise_std=0.0, opts=make_opts(), img_size=512, use_fine=True)
KEYS = ("img_coarse", "depth_rnd", "sil_coarse", "feat_rnd", "xyz_canonical_vis", "frame_cyc_dis")


def render(rr, mode):
    moda_amd.set_precision(mode)
    try:
        return moda_amd.render_rays(models, emb, rr, **kw)
    finally:
        moda_amd.set_precision("fp32")


ref = render(rays, "fp32")
torch.cuda.synchronize()
for feat, pre, prewarp, x3warp in (("bf16x3", "bf16x3", "", ""), ("fp16", "bf16x3", "", ""), ("fp16", "bf16x3", "", "fp16"), ("bf16x3", "fp16", "", ""),
                                   ("fp16", "fp16", "", "")):
    if True:
        R.FP16_FEAT_PRECISION, R.FP16_PREPASS_PRECISION, R.FP16_PREPASS_WARP = "bf16x3", pre, prewarp
        R.FP16_FEAT_UNCONSUMED_PRECISION, R.FP16_X3_PREPASS_WARP = feat, x3warp
        prewarp = prewarp or x3warp
        r = render(rays, "fp16")
        moda_amd.overflow.check()
        line = []
        for k in KEYS:
            if k not in r or k not in ref:
                continue
            a, b = r[k].float(), ref[k].float()
            d, mx = (a - b).abs(), b.abs().max().clamp_min(1e-30)
            line.append(f"{k} {float(d.max() / mx):.1e}/{float((d / (1e-4 * b.abs() + 1e-5 * mx)).max()):.2f}")
        for _ in range(2):
            render(rays_big, "fp16")
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(4):
            render(rays_big, "fp16")
        torch.cuda.synchronize(); dt = (time.time() - t0) / 4
        print(f"feat {feat:7s} pre-pass {pre:7s} (its warp {prewarp or pre:7s}): {65536 / dt / 1e6:.3f} M rays/s ({dt * 1e3:.1f} ms) | rel / per-element: " + "  ".join(line), flush=True)
moda_amd.set_precision("bf16")
for _ in range(2):
    moda_amd.render_rays(models, emb, rays_big, **kw)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(4):
    moda_amd.render_rays(models, emb, rays_big, **kw)
torch.cuda.synchronize(); print(f"bf16 mode: {65536 / ((time.time() - t0) / 4) / 1e6:.3f} M rays/s")
