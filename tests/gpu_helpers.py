"""Build moda_amd modules / `models` dicts on the GPU from the deterministic synth parameters."""
import types

import numpy as np
import torch

import moda_amd
from moda_amd import synth

DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def nerf_from_params(p, **kw):
    m = moda_amd.NeRF(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    return m.to(DEV).eval()


def make_models(seed, B, with_skin=True, with_feat=False, with_vis=False, alpha=10.0, perturb_bones=False, beta=0.1,
                with_dis=False):
    mp = synth.make_models(seed, B=B, with_skin=with_skin, with_feat=with_feat, with_vis=with_vis,
                           perturb_bones=perturb_bones, beta=beta, with_dis=with_dis)
    models = {"coarse": nerf_from_params(mp["coarse"], in_channels_xyz=63, in_channels_dir=27 + 64, init_beta=beta)}
    if B > 0:
        models["bones"] = torch.nn.Parameter(T(mp["bones_rst"]))
        models["bones_rst"] = T(mp["bones_rst"])
        models["skin_aux"] = T(mp["skin_aux"])
        rpc = torch.nn.Embedding(1, 128).to(DEV)
        if with_skin:
            models["nerf_skin"] = nerf_from_params(mp["nerf_skin"], D=5, W=64, in_channels_xyz=63 + 128,
                                                   in_channels_dir=0, out_channels=B, raw_feat=True,
                                                   in_channels_code=128)
            rpc.weight.data = T(mp["rest_pose_code"])
        models["rest_pose_code"] = rpc
    if with_dis:
        models["nerf_dis"] = nerf_from_params(mp["nerf_dis"], D=5, W=128, in_channels_xyz=63 + 128, in_channels_dir=0,
                                              out_channels=3, raw_feat=True, in_channels_code=128)
    if with_feat:
        models["nerf_feat"] = nerf_from_params(mp["nerf_feat"], D=5, W=128, in_channels_xyz=63, in_channels_dir=0,
                                               out_channels=16, raw_feat=True, init_beta=1.0)
    if with_vis:
        models["nerf_vis"] = nerf_from_params(mp["nerf_vis"], D=5, W=64, in_channels_xyz=63, in_channels_dir=0,
                                              out_channels=1, raw_feat=True)
    emb = {"xyz": moda_amd.Embedding(3, 10, alpha=alpha), "dir": moda_amd.Embedding(3, 4, alpha=alpha)}
    return models, emb


def make_opts(**kw):
    o = dict(dist_corresp=False, lbs=False, neudbs=True, symm_shape=False, scale_rgb=1.3, rgb_filter=False,
             use_corresp=False, use_corr=False, use_ot=False, s3im_loss=False)
    o.update(kw)
    return types.SimpleNamespace(**o)


def rays_to_gpu(rays):
    return {k: T(v) for k, v in rays.items()}


def unc_models(seed, B=25):
    """The G19 scene on the GPU: coarse net with the 128-wide appearance code, skin net, uncertainty head
    (moda.py:263-273, 456-464)."""
    from helpers import unc_scene_params
    mp = unc_scene_params(seed)
    models, emb = make_models(seed, B, with_skin=True, perturb_bones=True)
    models["coarse"] = nerf_from_params(mp["coarse"], in_channels_xyz=63, in_channels_dir=27 + 64 + 128, init_beta=0.1)
    unc = moda_amd.NeRFUnc(in_channels_xyz=63, D=8, W=256, out_channels=1, in_channels_dir=32, raw_feat=True, init_beta=1.)
    unc.load_state_dict({k: torch.from_numpy(v) for k, v in mp["nerf_unc"].items()})
    models["nerf_unc"] = unc.to(DEV).eval()
    return models, emb
