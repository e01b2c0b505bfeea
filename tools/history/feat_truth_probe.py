"""Where does nerf_feat's first-layer weight gradient lose accuracy?  G11 scene, loss = the heads' terms only; HIP path (fp32 training
precision) against oracle/torch_ref.py evaluated in float64 on the CPU, every nerf_feat / nerf_vis parameter gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import moda_amd  # noqa
from moda_amd import synth
from oracle import torch_ref as tr
from helpers import golden, rel_l2
from gpu_helpers import T, make_models, make_opts, rays_to_gpu
import test_torch_ref as ttr

TC = torch.from_numpy
use_ot = "--softmax" not in sys.argv
mode = "train_ot" if use_ot else "train_softmax"
g = golden("g11_heads_" + mode)
N, S, B = 48, 12, 25
TERMS = ("pts_pred", "feat_err", "proj_err", "vis_loss", "frnd_loss_samp")
only = [a[7:] for a in sys.argv if a.startswith("--term=")]
if only:
    TERMS = tuple(only)

# ---- float64 truth on the CPU
m = ttr.torch_scene_heads(11, B, requires_grad=False)
m = {k: ({kk: vv.double().requires_grad_(True) for kk, vv in v.items()} if isinstance(v, dict) else v.double().requires_grad_(True)) for k, v in m.items()}
rays_c = {k: v.double() for k, v in ttr.g11_rays().items()}
res_c = tr.render_rays(m, rays_c, S)
heads_c = tr.feature_heads(m, rays_c, res_c, ttr.G11_BOUND, use_ot, 512, feat_noise=TC(g["rng_randn_like"]).double(),
                           vis_neg_rand=TC(g["rng_rand"]).double(), training=True)
loss_c = 0
for k in TERMS:
    c = TC(synth.normal(11, "g11/c/" + k, tuple(heads_c[k].shape) or (1,))).reshape(heads_c[k].shape).double()
    loss_c = loss_c + (c * heads_c[k]).sum()
loss_c.backward()

# ---- HIP path
models, emb = make_models(11, B, with_skin=True, with_feat=True, with_vis=True, perturb_bones=True)
for mm in models.values():
    if isinstance(mm, torch.nn.Module):
        mm.train()
rays = rays_to_gpu(synth.make_rays(11, N, B, rays_per_frame=8))
rays.update(rays_to_gpu(synth.make_corresp_rays(11, N, B, rays_per_frame=8)))
rays.update(rays_to_gpu(synth.make_feat_rays(11, N, rays_per_frame=8)))
rng = {"feat_noise": T(g["rng_randn_like"]), "vis_neg_rand": TC(g["rng_rand"])}
with torch.enable_grad():
    res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, img_size=512, obj_bound=ttr.G11_BOUND,
                               opts=make_opts(dist_corresp=True, use_corresp=True, use_ot=use_ot), rng=rng)
loss = 0
for k in TERMS:
    c = T(synth.normal(11, "g11/c/" + k, tuple(res[k].shape) or (1,))).reshape(res[k].shape)
    loss = loss + (c * res[k]).sum()
    print(f"forward {k}: rel_l2 vs float64 {rel_l2(res[k].detach().cpu().numpy(), heads_c[k].detach().numpy()):.2e}")
loss.backward()
print("terms", TERMS, "loss", float(loss), float(loss_c))
for mn in ("nerf_feat", "nerf_vis"):
    for pn, p in models[mn].named_parameters():
        gc = m[mn][pn].grad
        if p.grad is None or gc is None:
            continue
        a, b = p.grad.cpu().numpy().astype(np.float64), gc.numpy()
        line = f"{mn}.{pn:28s} rel_l2 {rel_l2(a, b):.2e}  |g| {np.linalg.norm(b):.3e}"
        if a.ndim == 2 and a.shape[1] in (63, 191):      # per PE column group: xyz, then sin/cos per frequency
            cols = [rel_l2(a[:, :3], b[:, :3])] + [rel_l2(a[:, 3 + 6 * f:9 + 6 * f], b[:, 3 + 6 * f:9 + 6 * f]) for f in range(10)]
            line += "  per frequency: " + " ".join(f"{c:.0e}" for c in cols)
        print(line)
