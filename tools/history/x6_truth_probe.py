"""Which configuration makes the bf16x6 training precision lose 1e-3 against the float64 truth (seen on G25: 36 bones + symm_shape,
32 samples)?  Runs render_rays + backward for several (B, S) in fp32 and bf16x6 against oracle/torch_ref.py in float64."""
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
from helpers import rel_l2
from gpu_helpers import T, make_models, make_opts, rays_to_gpu
import test_torch_ref as ttr

LEAVES = ("rays_o", "rays_d", "bone_rts", "time_embedded", "env_code")


def truth(seed, N, S, B, rpf):
    m = ttr.torch_scene(seed, B, True, perturb_bones=True, requires_grad=True, dtype=torch.float64)
    rays = {k: torch.from_numpy(v).double() for k, v in synth.make_rays(seed, N, B, rays_per_frame=rpf).items()}
    for k in LEAVES:
        rays[k].requires_grad_(True)
    res = tr.render_rays(m, rays, S)
    loss = 0
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "frame_cyc_dis"):
        loss = loss + (torch.from_numpy(synth.normal(seed, "p/c/" + k, tuple(res[k].shape))).double() * res[k]).sum()
    loss.backward()
    g = {"d_" + k: rays[k].grad.numpy() for k in LEAVES}
    g.update({f"d_{mn}.{pn}": p.grad.numpy() for mn in ("coarse", "nerf_skin") for pn, p in m[mn].items() if p.grad is not None})
    g.update({"d_bones_rst": m["bones_rst"].grad.numpy(), "d_skin_aux": m["skin_aux"].grad.numpy(), "d_rest_pose_code": m["rest_pose_code"].grad.numpy()})
    return g


def hip(seed, N, S, B, rpf, prec):
    models, emb = make_models(seed, B, with_skin=True, perturb_bones=True)
    for mm in models.values():
        if isinstance(mm, torch.nn.Module):
            mm.train()
    models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
    models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
    rays = rays_to_gpu(synth.make_rays(seed, N, B, rays_per_frame=rpf))
    for k in LEAVES:
        rays[k].requires_grad_(True)
    moda_amd.set_train_precision(prec)
    try:
        res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, img_size=512, opts=make_opts())
        loss = 0
        for k in ("img_coarse", "depth_rnd", "sil_coarse", "frame_cyc_dis"):
            loss = loss + (T(synth.normal(seed, "p/c/" + k, tuple(res[k].shape))) * res[k]).sum()
        loss.backward()
    finally:
        moda_amd.set_train_precision("fp32")
    g = {"d_" + k: rays[k].grad.cpu().numpy() for k in LEAVES}
    g.update({f"d_{mn}.{pn}": p.grad.cpu().numpy() for mn in ("coarse", "nerf_skin") for pn, p in models[mn].named_parameters() if p.grad is not None})
    g.update({"d_bones_rst": models["bones_rst"].grad.cpu().numpy(), "d_skin_aux": models["skin_aux"].grad.cpu().numpy(),
              "d_rest_pose_code": models["rest_pose_code"].weight.grad.cpu().numpy()})
    return g


for (seed, N, S, B, rpf) in () if "--rays" in sys.argv else ((25, 64, 32, 36, 16), (25, 64, 32, 25, 16), (25, 64, 12, 36, 16), (9, 48, 12, 25, 8), (25, 64, 32, 32, 16), (25, 64, 32, 40, 16)):
    t = truth(seed, N, S, B, rpf)
    for prec in ("fp32", "bf16x6"):
        g = hip(seed, N, S, B, rpf, prec)
        errs = sorted(((rel_l2(g[k], t[k]), k) for k in t if k in g), reverse=True)
        print(f"seed {seed} N {N} S {S} B {B} [{prec:6s}] median {np.median([e for e, _ in errs]):.1e} worst: " +
              ", ".join(f"{k}={e:.1e}" for e, k in errs[:5]), flush=True)

if "--rays" in sys.argv:
    seed, N, S, B, rpf = 25, 64, 32, 36, 16
    t = truth(seed, N, S, B, rpf)
    a, b = hip(seed, N, S, B, rpf, "fp32"), hip(seed, N, S, B, rpf, "bf16x6")
    for k in ("d_rays_o", "d_rays_d", "d_bone_rts", "d_time_embedded"):
        sc = np.abs(t[k]).max()
        ea = np.abs(a[k] - t[k]).reshape(N, -1).max(1) / sc
        eb = np.abs(b[k] - t[k]).reshape(N, -1).max(1) / sc
        print(k, "per-ray max err / max|g|: fp32 worst rays", np.argsort(-ea)[:4], np.sort(ea)[::-1][:4], "| bf16x6 worst rays", np.argsort(-eb)[:6], np.sort(eb)[::-1][:6])
