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


def test_reference_written_checkpoint_renders_like_the_reference(tmp_path):
    """G20: the key -> shape map is `state_dict()` of the reference's own classes under the reference model's attribute
    names (tests/golden/gen_golden.py::_RefModel, moda.py:186-465) and `vars_*.npy` holds the bytes np.save wrote for a
    `latest_vars` dict as train_utils.save_network does (:298-304).  A params_*.pth with those keys (DDP 'module.' prefix,
    as the trainer saves) must load through moda_amd.checkpoint with nothing left over, the per-frame feeders built from it
    must reproduce the reference's FrameCode / DQ_RTHead outputs, and render_rays must reproduce the reference's render."""
    from helpers import golden, checkpoint_states, elem_err
    g = golden("g20_checkpoint")
    sd = checkpoint_states(g)
    assert len(sd) == 144 and "nerf_body_rts.0.basis_mlp.weight" in sd and "nerf_unc.dir_encoding.0.weight" in sd
    torch.save({"module." + k: torch.from_numpy(v) for k, v in sd.items()}, tmp_path / "params_7.pth")
    (tmp_path / "vars_7.npy").write_bytes(g["vars_npy_bytes"].tobytes())

    st = CK.load_params(str(tmp_path / "params_7.pth"))
    assert set(st) == set(sd)
    lv = CK.load_vars(str(tmp_path / "vars_7.npy"))
    assert lv["obj_bound"].shape == (3,) and abs(float(lv["obj_bound"][0]) - 0.27) < 1e-12 and lv["rtk"].shape == (19, 4, 4)
    offset = [0, 7, 19]
    models, emb, extras = CK.build_models(st, device=DEV, data_offset=offset, num_freqs=10)
    assert set(models) >= {"coarse", "nerf_skin", "nerf_feat", "nerf_vis", "nerf_unc", "bones", "bones_rst", "skin_aux",
                           "rest_pose_code"}
    assert isinstance(models["nerf_unc"], moda_amd.NeRFUnc) and models["nerf_unc"].in_channels_dir == 32
    assert set(extras) >= {"pose_code", "env_code", "nerf_body_rts", "near_far", "vid_code"}
    # every tensor of the checkpoint ended up in some module / tensor with the same values
    loaded = {}
    for name, key in (("coarse", "nerf_coarse"), ("nerf_skin", "nerf_skin"), ("nerf_feat", "nerf_feat"), ("nerf_vis", "nerf_vis"),
                      ("nerf_unc", "nerf_unc")):
        loaded.update({f"{key}.{k}": v for k, v in models[name].state_dict().items()})
    loaded.update({f"pose_code.{k}": v for k, v in extras["pose_code"].state_dict().items()})
    loaded.update({f"env_code.{k}": v for k, v in extras["env_code"].state_dict().items()})
    loaded.update({f"nerf_body_rts.{k}": v for k, v in extras["nerf_body_rts"].state_dict().items()})
    loaded.update({"bones": models["bones"], "skin_aux": models["skin_aux"], "rest_pose_code.weight": models["rest_pose_code"].weight,
                   "near_far": extras["near_far"], "vid_code.weight": extras["vid_code"].weight, "alpha": torch.tensor([emb["xyz"].alpha])})
    assert set(loaded) == set(sd), set(sd) ^ set(loaded)
    for k, v in sd.items():
        assert np.array_equal(loaded[k].detach().cpu().numpy().reshape(v.shape), v), k

    fid = torch.tensor([0, 3, 6, 7, 12, 18, 1, 9], device=DEV)
    N, S, F = 32, 12, 8
    with torch.no_grad():
        bone_rts = extras["nerf_body_rts"](fid).reshape(F, -1)
        tcode = extras["pose_code"](fid).reshape(F, -1)
        env = extras["env_code"](fid).reshape(F, -1)
    for got, key in ((bone_rts, "bone_rts"), (tcode, "time_embedded"), (env, "env_code")):
        e = rel_err(got.cpu().numpy(), g[key])
        assert e < 1e-5, (key, e)
    rays = rays_to_gpu(synth.make_rays(20, N, 0, rays_per_frame=N // F))
    rep = lambda t: t[:, None].repeat(1, N // F, 1).reshape(N, -1)
    rays["bone_rts"], rays["time_embedded"], rays["env_code"] = rep(bone_rts), rep(tcode), rep(env)
    rays.update(rays_to_gpu(synth.make_unc_rays(20, N, N // F)))
    vid = torch.tensor([0 if f < offset[1] else 1 for f in fid.tolist()], device=DEV)
    rays["vid_code"] = rep(extras["vid_code"](vid))
    assert rel_err(rays["vid_code"].detach().cpu().numpy(), g["vid_code"]) < 1e-7
    with torch.no_grad():
        res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512,
                                   render_vis=True, obj_bound=np.asarray([0.3, 0.3, 0.3]))
    for k in ("img_coarse", "sil_coarse", "depth_rnd", "xyz_canonical_vis", "frame_cyc_dis", "vis_pred", "unc_pred"):
        e = rel_err(res[k].cpu().numpy(), g["render_" + k])
        assert e < 1e-4, (k, e)
        assert elem_err(res[k].cpu().numpy(), g["render_" + k]) < 1, (k, elem_err(res[k].cpu().numpy(), g["render_" + k]))


def test_frame_code_width_is_validated():
    """FrameCode's input width is n_vids * (1 + 2 F) (nerf.py:359-361): a data_offset that does not match the checkpoint
    is refused instead of silently mis-slicing the basis."""
    w = torch.zeros(128, 42)
    st = {"pose_code.basis_mlp.weight": w, "pose_code.basis_mlp.bias": torch.zeros(128)}
    mp_ = synth.make_models(31, B=0)
    st.update({f"nerf_coarse.{k}": torch.from_numpy(v) for k, v in mp_["coarse"].items()})
    with pytest.raises(ValueError):
        CK.build_models(st, device=DEV, data_offset=[0, 5, 9, 30], num_freqs=10)     # 3 videos: 42 is not 3 * (1 + 2F)
    CK.build_models(st, device=DEV, data_offset=[0, 5, 30], num_freqs=10)
