"""GPU (-m gpu): a MoDA checkpoint pair (params_*.pth + vars_*.npy, nnutils/train_utils.py:292-306) written under the
reference model's key names loads into moda_amd and renders the same images as modules built directly."""
import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import moda_amd
    from moda_amd import synth, checkpoint as CK
    from gpu_helpers import T, DEV, make_models, make_opts, rays_to_gpu


def test_reference_checkpoint_layout_loads_and_renders(tmp_path):
    B, C = 25, 128
    mp_ = synth.make_models(31, B=B, with_skin=True, with_feat=True, with_vis=True, perturb_bones=True)
    states = {}
    for attr, key in (("nerf_coarse", "coarse"), ("nerf_skin", "nerf_skin"), ("nerf_feat", "nerf_feat"), ("nerf_vis", "nerf_vis")):
        for k, v in mp_[key].items():
            states[f"module.{attr}.{k}"] = torch.from_numpy(v)
    states["module.bones"] = torch.from_numpy(mp_["bones_rst"])
    states["module.skin_aux"] = torch.from_numpy(mp_["skin_aux"])
    states["module.rest_pose_code.weight"] = torch.from_numpy(mp_["rest_pose_code"])
    states["module.alpha"] = torch.tensor([10.0])
    states["module.near_far"] = torch.from_numpy(np.tile(np.asarray([[0.1, 0.5]], np.float32), (50, 1)))
    fw, fb = synth.linear_init(31, "ck/pose", C, 13)
    states["module.pose_code.basis_mlp.weight"], states["module.pose_code.basis_mlp.bias"] = torch.from_numpy(fw), torch.from_numpy(fb)
    states["module.nerf_body_rts.0.basis_mlp.weight"], states["module.nerf_body_rts.0.basis_mlp.bias"] = torch.from_numpy(fw), torch.from_numpy(fb)
    head = synth.nerf_params(31, "ck/head", D=8, W=256, in_channels_xyz=C, in_channels_dir=0, out_channels=7 * B)
    for k, v in head.items():
        states[f"module.nerf_body_rts.1.{k}"] = torch.from_numpy(v)
    torch.save(states, tmp_path / "params_latest.pth")
    np.save(tmp_path / "vars_latest.npy", {"obj_bound": np.asarray(0.2), "rtk": np.zeros((50, 4, 4))})

    st = CK.load_params(str(tmp_path / "params_latest.pth"))
    assert not any(k.startswith("module.") for k in st)
    lv = CK.load_vars(str(tmp_path / "vars_latest.npy"))
    assert lv["obj_bound"].shape == (3,)
    models, emb, extras = CK.build_models(st, device=DEV, data_offset=[0, 50], num_freqs=10)
    assert set(models) >= {"coarse", "nerf_skin", "nerf_feat", "nerf_vis", "bones", "bones_rst", "skin_aux", "rest_pose_code"}
    assert models["nerf_skin"].W == 64 and models["nerf_skin"].in_channels_xyz == 191 and models["nerf_feat"].out_channels == 16
    assert set(extras) >= {"pose_code", "nerf_body_rts", "near_far"}
    dq = extras["nerf_body_rts"](torch.tensor([3, 7], device=DEV))
    assert tuple(dq.shape) == (2, 1, 8 * B)

    ref_models, ref_emb = make_models(31, B, with_skin=True, with_feat=True, with_vis=True, perturb_bones=True)
    rays = rays_to_gpu(synth.make_rays(31, 64, B, rays_per_frame=16))
    with torch.no_grad():
        a = moda_amd.render_rays(models, emb, rays, N_samples=16, noise_std=0.0, opts=make_opts(), img_size=512, render_vis=True,
                                 obj_bound=lv["obj_bound"])
        b = moda_amd.render_rays(ref_models, ref_emb, rays, N_samples=16, noise_std=0.0, opts=make_opts(), img_size=512,
                                 render_vis=True, obj_bound=lv["obj_bound"])
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "frame_cyc_dis", "vis_pred"):
        assert torch.equal(a[k], b[k]), k
