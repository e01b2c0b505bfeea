"""GPU (-m gpu): the per-frame feeders before the path (moda_amd/feeders.py) against the reference's
outputs and autograd (tests/golden/g12_feeders.npz)."""
import types

import numpy as np
import pytest
import torch

from helpers import golden, rel_err

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import moda_amd
    from moda_amd import synth, feeders as FD
    from gpu_helpers import T, DEV

G12 = dict(F=6, ns=11, B=5, code=32, vid_offset=[0, 40, 100], n_freq=6)


def np_(t):
    return t.detach().cpu().numpy()


def test_raycast_matches_reference():
    g = golden("g12_feeders")
    F_, ns = G12["F"], G12["ns"]
    cam = {k: T(v).requires_grad_(k != "near_far") for k, v in synth.make_cameras(12, F_).items()}
    xys = T(synth.uniform(12, "g12/xys", (F_, ns, 2)) * np.float32(512))
    for tag, nf in (("nf", cam["near_far"]), ("auto", None)):
        rays = FD.raycast(xys, cam["Rmat"], cam["Tmat"], cam["Kinv"], nf)
        assert rays["nsample"] == ns and rays["bs"] == F_
        for k in ("rays_o", "rays_d", "near", "far", "rtk_vec"):
            assert rays[k].shape == g[f"raycast_{tag}_{k}"].shape, k
            assert rel_err(np_(rays[k]), g[f"raycast_{tag}_{k}"]) < 1e-5, (tag, k)
        if tag == "nf":
            loss = (T(synth.normal(12, "g12/c/d", (F_, ns, 3))) * rays["rays_d"]).sum() \
                + (T(synth.normal(12, "g12/c/o", (F_, ns, 3))) * rays["rays_o"]).sum()
            loss.backward()
            for k in ("Rmat", "Tmat", "Kinv"):
                assert rel_err(np_(cam[k].grad), g["raycast_d_" + k]) < 1e-4, k
    chunk = FD.chunk_rays(rays, 5, 20)
    assert chunk["rays_d"].shape == (20, 3) and torch.equal(chunk["rays_d"], rays["rays_d"].reshape(-1, 3)[5:25])


def test_frame_code_matches_reference():
    g = golden("g12_feeders")
    C = G12["code"]
    fc = FD.FrameCode(G12["n_freq"], C, np.asarray(G12["vid_offset"])).to(DEV)
    w, b = synth.linear_init(12, "g12/fc", C, fc.basis_mlp.in_features)
    fc.basis_mlp.weight.data, fc.basis_mlp.bias.data = T(w), T(b)
    fid = torch.tensor([0, 3, 39, 40, 41, 77, 99, 12], device=DEV)
    code = fc(fid)
    (T(synth.normal(12, "g12/c/code", tuple(code.shape))) * code).sum().backward()
    assert rel_err(np_(code), g["framecode"]) < 1e-5
    assert rel_err(np_(fc.basis_mlp.weight.grad), g["framecode_d_weight"]) < 1e-4


def test_dq_rthead_matches_reference():
    g = golden("g12_feeders")
    B, C = G12["B"], G12["code"]
    kw = dict(D=8, W=64, in_channels_xyz=C, in_channels_dir=0, out_channels=7 * B, raw_feat=True)
    head = FD.DQ_RTHead(use_quat=True, **kw)
    pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_params(12, "g12/head", **pk).items()})
    head = head.to(DEV).train()
    x = T(synth.normal(12, "g12/x", (8, C))).requires_grad_(True)
    dq = head(x)
    (T(synth.normal(12, "g12/c/dq", tuple(dq.shape))) * dq).sum().backward()
    assert dq.shape == g["rthead"].shape and rel_err(np_(dq), g["rthead"]) < 1e-5
    assert rel_err(np_(x.grad), g["rthead_d_x"]) < 2e-4
    assert rel_err(np_(head.rgb[0].weight.grad), g["rthead_d_rgb"]) < 2e-4
    assert rel_err(np_(head.xyz_encoding_5[0].weight.grad), g["rthead_d_l5"]) < 2e-4
    with torch.no_grad():
        assert rel_err(np_(head.eval()(x.detach())), g["rthead"]) < 1e-5          # no-grad route


def test_rest_pose_correction_matches_reference():
    g = golden("g12_feeders")
    B = G12["B"]
    fw = T(synth.frame_dual_quats(12, "g12/fw", 7, B)).requires_grad_(True)
    rst = T(synth.frame_dual_quats(12, "g12/rst", 1, B)).requires_grad_(True)
    delta = FD.correct_rest_pose(types.SimpleNamespace(num_bones=B), fw[:, None], rst, True)
    (T(synth.normal(12, "g12/c/delta", tuple(delta.shape))) * delta).sum().backward()
    assert delta.shape == g["rest_delta"].shape and rel_err(np_(delta), g["rest_delta"]) < 1e-5
    assert rel_err(np_(fw.grad), g["rest_d_fw"]) < 1e-4
    assert rel_err(np_(rst.grad), g["rest_d_rst"]) < 1e-4


def test_feeders_drive_render_rays_end_to_end():
    """cameras + frame ids -> raycast -> FrameCode / DQ_RTHead -> update_rays -> render_rays: the frame-level inputs a MoDA
    training step builds (moda.py:1281-1327), gradients reaching the camera and the pose head."""
    from gpu_helpers import make_models, make_opts
    F_, ns, B, S = 4, 32, 25, 16
    models, emb = make_models(13, B, with_skin=True)
    for m in models.values():
        if isinstance(m, torch.nn.Module):
            m.train()
    cam = {k: T(v).requires_grad_(k != "near_far") for k, v in synth.make_cameras(13, F_).items()}
    xys = T(synth.uniform(13, "e2e/xys", (F_, ns, 2)) * np.float32(512))
    model = types.SimpleNamespace()
    model.pose_code = FD.FrameCode(6, 128, np.asarray([0, 50])).to(DEV)
    model.env_code = FD.FrameCode(6, 64, np.asarray([0, 50])).to(DEV)
    head = FD.DQ_RTHead(use_quat=True, in_channels_xyz=128, in_channels_dir=0, out_channels=7 * B, raw_feat=True).to(DEV)
    with torch.no_grad():   # start near the identity transform, as the reference's zero-bias init does in spirit
        head.rgb[0].weight.mul_(0.05)
        head.rgb[0].bias.copy_(torch.tensor([0, 0, 0, 1, 0, 0, 0.0], device=DEV).repeat(B))
    model.nerf_body_rts = torch.nn.Sequential(model.pose_code, head)
    rays = FD.raycast(xys, cam["Rmat"], cam["Tmat"], cam["Kinv"], None)
    rays["near"] = torch.full_like(rays["near"], 0.6)
    rays["far"] = torch.full_like(rays["far"], 1.4)
    rays = FD.update_rays(model, rays, False, torch.arange(F_, device=DEV) * 7)
    flat = {k: v.reshape(F_ * ns, -1) for k, v in rays.items() if torch.is_tensor(v)}
    res = moda_amd.render_rays(models, emb, flat, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
    loss = res["img_coarse"].pow(2).mean() + res["frame_cyc_dis"].mean()
    loss.backward()
    for t in (cam["Rmat"], cam["Tmat"], cam["Kinv"], head.rgb[0].weight, model.pose_code.basis_mlp.weight,
              model.env_code.basis_mlp.weight):
        assert t.grad is not None and torch.isfinite(t.grad).all() and float(t.grad.abs().max()) > 0


def test_line_batch_feeds_the_path(tmp_path):
    """Per-line pixel files (img2lines.py / LineDataset) -> set_input's pair-major layout -> one pixel per ray
    (obs_to_rays_line) -> the observed signals the loss block reads; features come out unit-norm, masks are
    mask * vis2d > 0 (moda.py:1329-1360)."""
    from moda_amd import pixel_lines as PL
    from test_pixel_lines import frame_pair
    H, bs = 8, 3
    rng = np.random.default_rng(5)
    root = str(tmp_path / "seq")
    src = []
    for idt in range(bs):
        src.append(frame_pair(rng, H))
        PL.write_pair(PL.pair_dir(str(tmp_path), "seq", 1, idt), src[-1], H)
    ds = PL.LineDataset(root, bs + 1, H, dataid=0, rng=types.SimpleNamespace(choice=lambda c: c[0]))   # always dt = 1
    items = [ds[idt * H + (idt + 2)] for idt in range(bs)]                      # row idt+2 of pair idt
    batch = {k: np.concatenate([np.asarray(it[k]) for it in items], 0) for k in items[0]}
    inp = PL.set_input(batch, data_offset=[0, bs + 1], img_size=H, device=DEV)
    assert inp['imgs'].shape == (2 * bs, 3, H, 1) and inp['dp_feats'].shape == (2 * bs, 16, H, 1)
    assert torch.allclose(inp['dp_feats'][..., 0].norm(dim=1), torch.ones(2 * bs, H, device=DEV), atol=1e-5)
    for i in range(bs):
        row = i + 2
        assert np.allclose(np_(inp['imgs'][i, :, :, 0]), src[i]['img'][0, 0, :, row, :])            # frame t block
        assert np.allclose(np_(inp['imgs'][bs + i, :, :, 0]), src[i]['img'][0, 1, :, row, :])       # frame t+dt block
        m = (src[i]['mask'][0, 0, row] * src[i]['vis2d'][0, 0, row] > 0).astype(np.float32)
        assert np.array_equal(np_(inp['masks'][i, 0, :, 0]), m)
    assert inp['frameid'].tolist() == [0, 1, 2, 1, 2, 3] and inp['lineid'].tolist() == [2, 3, 4, 2, 3, 4]
    assert inp['errid'].tolist() == [f * H + l for f, l in zip(inp['frameid'].tolist(), inp['lineid'].tolist())]
    # one pixel per ray, as sample_pxs reshapes for line loading (moda.py:1180-1190)
    ns = 5
    cols, xys = FD.sample_xy(H, 2 * bs, ns, DEV, return_all=False, lineid=inp['lineid'])
    assert torch.equal(xys[..., 1], inp['lineid'][:, None].expand(-1, ns)) and int(cols.max()) < H
    bm = torch.arange(2 * bs, device=DEV)[:, None].expand(-1, ns).reshape(-1)
    rays = PL.obs_to_rays_line({}, cols.reshape(-1, 1), inp['imgs'], inp['masks'], inp['vis2d'], inp['flow'], inp['occ'],
                               inp['dp_feats'], bm)
    assert rays['img_at_samp'].shape == (2 * bs * ns, 1, 3) and rays['feats_at_samp'].shape == (2 * bs * ns, 1, 16)
    j = 2 * ns + 1                                                                                   # frame 2, 2nd sample
    assert torch.equal(rays['img_at_samp'][j, 0], inp['imgs'][2, :, cols[2, 1], 0])


def _to_frames(rays_np, k):
    """Per-ray dict of synth.make_rays -> frame-grouped layout: every FRAME_KEYS tensor keeps one row per frame."""
    from moda_amd.rendering import FRAME_KEYS
    out = {key: T(v[::k].copy()) if key in FRAME_KEYS else T(v) for key, v in rays_np.items()}
    out['rays_per_frame'] = k
    return out


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("S,k", [(256, 8), (128, 4), (50, 3)])
def test_frame_grouped_layout_equals_per_ray_layout(S, k, precision):
    """The frame-grouped layout (one bone_rts / code row per frame, rays['rays_per_frame']) renders exactly what the
    reference's per-ray repeats render -- same kernels, the same arithmetic per ray -- on the multi-sample warp
    (S % 256, S % 128) and the generic one, with the paired-frame correspondence heads on."""
    from gpu_helpers import make_models, make_opts, rays_to_gpu
    N, B = 8 * k, 25
    models, emb = make_models(21, B, with_skin=True)
    rays_np = synth.make_rays(21, N, B, rays_per_frame=k)
    rays_np.update(synth.make_corresp_rays(21, N, B, rays_per_frame=k))
    opts = make_opts(dist_corresp=True)
    moda_amd.set_precision(precision)        # (bf16 / fp16: the one-kernel warps at S % 32 == 0, per-ray vs per-frame tables)
    try:
        _frame_layout_body(models, emb, rays_np, S, k, opts)
    finally:
        moda_amd.set_precision("fp32")


def _frame_layout_body(models, emb, rays_np, S, k, opts):
    from gpu_helpers import rays_to_gpu
    with torch.no_grad():
        ref = moda_amd.render_rays(models, emb, rays_to_gpu(rays_np), N_samples=S, noise_std=0.0, opts=opts, img_size=512)
        got = moda_amd.render_rays(models, emb, _to_frames(rays_np, k), N_samples=S, noise_std=0.0, opts=opts, img_size=512)
    assert set(ref) == set(got)
    for key in ref:
        assert torch.equal(ref[key], got[key]), key
    # chunking keeps whole frames and the layout
    ch = FD.chunk_rays(_to_frames(rays_np, k), 2 * k, 4 * k)
    with torch.no_grad():
        part = moda_amd.render_rays(models, emb, ch, N_samples=S, noise_std=0.0, opts=opts, img_size=512)
    assert torch.equal(part["img_coarse"], ref["img_coarse"][2 * k:6 * k])
    with pytest.raises(ValueError):
        FD.chunk_rays(_to_frames(rays_np, k), 1, k)


def test_frame_grouped_layout_gradients_are_frame_sums():
    """Training route: the gradient at a per-frame row equals the sum of the per-ray layout's gradients over that frame's
    rays (what autograd gives the reference through .repeat)."""
    from gpu_helpers import make_models, make_opts, rays_to_gpu
    N, B, S, k = 24, 25, 16, 6
    models, emb = make_models(22, B, with_skin=True)
    for m in models.values():
        if isinstance(m, torch.nn.Module):
            m.train()
    rays_np = synth.make_rays(22, N, B, rays_per_frame=k)
    grads = {}
    for tag, rays in (("ray", rays_to_gpu(rays_np)), ("frame", _to_frames(rays_np, k))):
        for key in ("bone_rts", "time_embedded", "env_code"):
            rays[key].requires_grad_(True)
        for m in models.values():
            if isinstance(m, torch.nn.Module):
                m.zero_grad()
        res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
        (res["img_coarse"].pow(2).sum() + res["frame_cyc_dis"].sum() + res["sil_coarse"].sum()).backward()
        grads[tag] = {key: rays[key].grad.clone() for key in ("bone_rts", "time_embedded", "env_code")}
        grads[tag]["w"] = models["nerf_skin"].xyz_encoding_1[0].weight.grad.clone()
    for key in ("bone_rts", "time_embedded", "env_code"):
        want = grads["ray"][key].view(N // k, k, -1).sum(1)
        assert grads["frame"][key].shape == want.shape
        assert rel_err(np_(grads["frame"][key]), np_(want)) < 1e-5, key
    assert rel_err(np_(grads["frame"]["w"]), np_(grads["ray"]["w"])) < 1e-5


def test_frame_grouped_layout_with_feature_heads_under_no_grad():
    """rays_per_frame > 1 together with 'feats_at_samp' (feature matching + keypoint reprojection -> forward_warp of ONE
    point per ray) and use_corresp without dist_corresp, evaluated without autograd: the per-frame bone_rts rows reach the
    warp as (frames, B, 8) with rays_per_set = k (round 1 raised 'dq: expected N sets' here)."""
    from gpu_helpers import make_models, make_opts, rays_to_gpu
    N, B, S, k = 32, 25, 16, 4
    models, emb = make_models(23, B, with_skin=True, with_feat=True, with_vis=True)
    rays_np = synth.make_rays(23, N, B, rays_per_frame=k)
    rays_np.update(synth.make_corresp_rays(23, N, B, rays_per_frame=k))
    rays_np.update(synth.make_feat_rays(23, N, rays_per_frame=k))
    bound = np.asarray([0.2, 0.2, 0.2], np.float32)
    fn = torch.from_numpy(synth.normal(23, "fn", (1, 8000, 3))).to(DEV)
    for opts in (make_opts(dist_corresp=True, use_corresp=True, use_ot=True), make_opts(use_corresp=True)):
        rn = dict(rays_np)
        if not opts.dist_corresp:     # the reprojected-point flow exists for the target frame only (rendering.py:485)
            rn.pop("rtk_vec_dentrg"), rn.pop("bone_rts_dentrg")
        with torch.no_grad():
            ref = moda_amd.render_rays(models, emb, rays_to_gpu(rn), N_samples=S, noise_std=0.0, opts=opts, img_size=512,
                                       obj_bound=bound, rng={"feat_noise": fn})
            got = moda_amd.render_rays(models, emb, _to_frames(rn, k), N_samples=S, noise_std=0.0, opts=opts, img_size=512,
                                       obj_bound=bound, rng={"feat_noise": fn})
        assert set(ref) == set(got) and "proj_err" in got and "flo_coarse" in got
        for key in ref:
            assert torch.equal(ref[key], got[key]), key
