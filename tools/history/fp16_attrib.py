"""Where the fp16 mode's position error comes from (development probe): per-element figure of xyz_canonical_vis / frame_cyc_dis
against the split-bf16 mode on 8192 rays of config 2, for the fused fp16 warp, the two-kernel route with an fp16 skin network
(exact VALU warp) and the fused bf16 warp; plus kernel times of the fp16 and bf16 modes."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import moda_amd
import moda_amd.rendering as R
from moda_amd import synth, nerf, _lib
from helpers import elem_err, rel_err
from gpu_helpers import make_models, make_opts, rays_to_gpu
np_ = lambda t: t.detach().cpu().numpy()
torch.set_grad_enabled(False)
N, S = 8192, 256
models, emb = make_models(0, 25)
rays = rays_to_gpu(synth.make_rays(1000, N, 25, rays_per_frame=256))
kw = dict(N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
moda_amd.set_precision("bf16x3")
ref = moda_amd.render_rays(models, emb, rays, **kw)
def show(tag, res):
    print(tag, " ".join(f"{k}: rel {rel_err(np_(res[k]), np_(ref[k])):.2e} elem {elem_err(np_(res[k]), np_(ref[k])):.2f}"
                        for k in ("img_coarse", "xyz_canonical_vis", "frame_cyc_dis")), flush=True)
moda_amd.set_precision("fp16")
show("fp16 fused warp        ", moda_amd.render_rays(models, emb, rays, **kw))
R.WARP_PRECISION["fp16"] = "bf16x3"
show("fp16 + x3 fused warp   ", moda_amd.render_rays(models, emb, rays, **kw))
R.WARP_PRECISION["fp16"] = "fp16"
R.FUSED_WARP = False
orig = nerf.default_precision
nerf.default_precision = lambda: "fp16"            # two-kernel route with the skin network in fp16
show("fp16 MLP + exact warp  ", moda_amd.render_rays(models, emb, rays, **kw))
nerf.default_precision = orig
show("x3 MLP + exact warp    ", moda_amd.render_rays(models, emb, rays, **kw))
R.FUSED_WARP = True
moda_amd.set_precision("bf16")
show("bf16 fused warp        ", moda_amd.render_rays(models, emb, rays, **kw))
# kernel times at full size
N = 65536
rays = rays_to_gpu(synth.make_rays(1000, N, 25, rays_per_frame=256))
for prec in ("bf16", "fp16", "fp16+x3warp", "bf16", "fp16", "fp16+x3warp", "bf16x3", "bf16x3+fusedwarp"):
    R.WARP_PRECISION["fp16"] = "bf16x3" if prec == "fp16+x3warp" else "fp16"
    R.WARP_PRECISION.pop("bf16x3", None)
    if prec == "bf16x3+fusedwarp":
        R.WARP_PRECISION["bf16x3"] = "bf16x3"
    moda_amd.set_precision(prec.split("+")[0])
    for _ in range(5):
        moda_amd.render_rays(models, emb, rays, **kw)
    torch.cuda.synchronize()
    _lib.PROFILE = {}
    t0 = time.time()
    for _ in range(20):
        moda_amd.render_rays(models, emb, rays, **kw)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 20
    prof, _lib.PROFILE = _lib.PROFILE, None
    print(prec, f"{dt*1e3:.2f} ms/step", {t: round(float(np.mean([s.elapsed_time(e) for s, e, _ in ev])), 3) for t, ev in prof.items()}, flush=True)
