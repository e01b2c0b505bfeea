"""GPU (-m gpu): canonical-grid queries of mesh extraction (moda_amd/mesh_queries.py) against the
reference's warp_bw / warp_fw and grid evaluations (tests/golden/g13_grid.npz)."""
import types

import numpy as np
import pytest
import torch

from helpers import golden, rel_err

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import moda_amd
    from moda_amd import synth, feeders as FD, mesh_queries as MQ
    from gpu_helpers import T, DEV, make_models, nerf_from_params

G13 = dict(B=25, P=200, grid=6, embedid=3, vid_offset=[0, 50], code=128)


def np_(t):
    return t.detach().cpu().numpy()


def build_model():
    B, C = G13["B"], G13["code"]
    hk = dict(D=8, W=256, in_channels_xyz=C, in_channels_dir=0, out_channels=7 * B)
    head_p = synth.nerf_params(13, "g13/head", **hk)
    head_p["rgb.0.weight"] = head_p["rgb.0.weight"] * np.float32(0.05)
    head_p["rgb.0.bias"] = np.tile(np.asarray([0, 0, 0, 1, 0, 0, 0], np.float32), B) + np.float32(0.1) * synth.normal(13, "g13/head/b", (7 * B,))
    fw, fb = synth.linear_init(13, "g13/pose", C, 2 * (1 + 2 * 6))
    models, emb = make_models(13, B, with_skin=True, with_vis=True, perturb_bones=True, with_dis=True)
    model = types.SimpleNamespace(device=DEV)
    model.embedding_xyz = emb["xyz"]
    model.pose_code = FD.FrameCode(6, C, np.asarray(G13["vid_offset"])).to(DEV)
    model.pose_code.basis_mlp.weight.data, model.pose_code.basis_mlp.bias.data = T(fw[:, :model.pose_code.basis_mlp.in_features]), T(fb)
    head = FD.DQ_RTHead(use_quat=True, in_channels_xyz=C, in_channels_dir=0, out_channels=7 * B, raw_feat=True)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in head_p.items()})
    model.nerf_body_rts = torch.nn.Sequential(model.pose_code, head.to(DEV).eval())
    model.bones = models["bones_rst"]
    model.rest_pose_code = models["rest_pose_code"]
    model.nerf_skin = models["nerf_skin"]
    model.skin_aux = models["skin_aux"]
    model.nerf_dis = models["nerf_dis"]
    model.opts = types.SimpleNamespace(num_bones=B)
    return model, models, emb


def test_point_warps_match_reference():
    g = golden("g13_grid")
    model, _, _ = build_model()
    opts = types.SimpleNamespace(flowbw=False, lbs=False, neudbs=True, nerf_skin=True, nerf_dis=False, num_bones=G13["B"])
    pts = np.float32(0.15) * synth.normal(13, "g13/pts", (G13["P"], 3))
    bw, d1 = MQ.warp_bw(opts, model, {}, T(pts), G13["embedid"])
    fw, d2 = MQ.warp_fw(opts, model, {}, pts.copy(), G13["embedid"])
    assert rel_err(np_(bw), g["warp_bw"]) < 1e-4, rel_err(np_(bw), g["warp_bw"])
    assert isinstance(fw, np.ndarray) and rel_err(fw, g["warp_fw"]) < 1e-4, rel_err(fw, g["warp_fw"])
    assert rel_err(np_(d1["bones"])[0], g["warp_bw_bones"][0]) < 1e-5 and np.abs(g["warp_bw_bones"] - g["warp_bw_bones"][:1]).max() == 0
    assert rel_err(np_(d2["bones"])[0], g["warp_fw_bones"][0]) < 1e-5
    # with the residual displacement field (opts.nerf_dis): subtracted after the backward blend, added before the forward one
    opts_d = types.SimpleNamespace(**{**vars(opts), "nerf_dis": True})
    bwd, _ = MQ.warp_bw(opts_d, model, {}, T(pts), G13["embedid"])
    fwd, _ = MQ.warp_fw(opts_d, model, {}, pts.copy(), G13["embedid"])
    assert rel_err(np_(bwd), g["warp_bw_dis"]) < 1e-4 and rel_err(fwd, g["warp_fw_dis"]) < 1e-4
    assert np.abs(np_(bwd) - np_(bw)).max() > 1e-3
    # the pair is a cycle up to the skinning fields' mismatch: forward(backward(x)) stays near x
    back, _ = MQ.warp_fw(opts, model, {}, np_(bw), G13["embedid"])
    assert np.abs(back - pts).max() < 0.05


def test_volume_queries_match_reference():
    g = golden("g13_grid")
    _, models, emb = build_model()
    bound = np.asarray([0.2, 0.15, 0.25], np.float32)
    vol, vis = MQ.query_volume(models["coarse"], emb["xyz"], bound, G13["grid"], nerf_vis=models["nerf_vis"], precision="fp32")
    assert vol.shape == g["vol_sigma"].shape and rel_err(np_(vol), g["vol_sigma"]) < 1e-4
    assert rel_err(np_(vis), g["vol_vis"]) < 1e-4
    # throughput mode stays within the bf16 band, and a 64^3 lattice (the per-epoch extraction size) runs in one launch
    vol16, _ = MQ.query_volume(models["coarse"], emb["xyz"], bound, G13["grid"], precision="bf16")
    assert rel_err(np_(vol16), g["vol_sigma"]) < 3e-2
    big, bigv = MQ.query_volume(models["coarse"], emb["xyz"], bound, 64, nerf_vis=models["nerf_vis"], precision="bf16")
    assert big.shape == (64, 64, 64) and torch.isfinite(big).all() and float(bigv.min()) >= 0 and float(bigv.max()) <= 1
