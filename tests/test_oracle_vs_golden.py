"""CPU: pin the numpy oracle against golden vectors produced by the reference itself (tests/golden/gen_golden.py)."""
import numpy as np
import pytest

from moda_amd import synth
from oracle import moda_oracle as orc
from helpers import (E2E_CASES, e2e_random_inputs, golden, oracle_scene, rel_err, cast, elem_err, unc_scene_params,
                     checkpoint_states)

TOL = 2e-6  # fp32 oracle vs fp32 reference: same maths, different summation order


def test_g1_dual_quat():
    g = golden("g1_dual_quat")
    a = synth.normal(1, "g1/a", (37, 8))
    b = synth.normal(1, "g1/b", (37, 8))
    assert rel_err(orc.q_mul(a[:, :4], b[:, :4]), g["q_mul"]) < TOL
    assert rel_err(orc.dq_mul(a, b), g["dq_mul"]) < TOL
    assert rel_err(orc.dq_mul(a[None], b[None]), g["dq_mul_nd"]) < TOL
    assert rel_err(orc.dq_normalize(a), g["dq_normalize"]) < TOL
    assert rel_err(orc.dq_inverse(a), g["dq_inverse"]) < TOL
    assert np.array_equal(orc.dq_quaternion_conjugate(a), g["dq_qconj"])
    assert np.array_equal(orc.dq_combined_conjugate(a), g["dq_cconj"])
    assert rel_err(orc.q_normalize(a[:, :4]), g["q_normalize"]) < TOL
    # algebra: for UNIT dual quaternions (the only kind the path feeds it) dq * dq^-1 = identity
    u = synth.frame_dual_quats(1, "g1/unit", 5, 7).reshape(35, 8)
    ident = orc.dq_mul(u, orc.dq_inverse(u))
    assert np.abs(ident - np.asarray([1, 0, 0, 0, 0, 0, 0, 0], np.float32)).max() < 1e-5


def test_g2_embedding():
    g = golden("g2_embedding")
    x = synth.normal(2, "g2/x", (5, 7, 3))
    for alpha in (6.5, 10.0):
        assert rel_err(orc.embedding(x, 10, alpha), g[f"xyz_a{alpha}"]) < TOL
        assert rel_err(orc.embedding(x, 4, alpha), g[f"dir_a{alpha}"]) < TOL
    assert rel_err(orc.embedding(x, 10), g["xyz_default"]) < TOL
    e = orc.embedding(np.asarray([[0.1, 0.2, 0.3]], np.float32), 10)
    assert np.allclose(e[0, :7], [0.1, 0.2, 0.3, np.sin(0.1), np.sin(0.2), np.sin(0.3), np.cos(0.1)], atol=1e-6)


NERF_SHAPES = {
    "coarse": dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=27 + 64, out_channels=3, raw_feat=False),
    "skin": dict(D=5, W=64, in_channels_xyz=63 + 128, in_channels_dir=0, out_channels=25, raw_feat=True),
    "feat": dict(D=5, W=128, in_channels_xyz=63, in_channels_dir=0, out_channels=16, raw_feat=True),
    "vis": dict(D=5, W=64, in_channels_xyz=63, in_channels_dir=0, out_channels=1, raw_feat=True),
}


@pytest.mark.parametrize("name", list(NERF_SHAPES))
def test_g3_nerf(name):
    g = golden("g3_nerf")
    kw = NERF_SHAPES[name]
    pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
    p = synth.nerf_params(3, "g3/" + name, **pk)
    x = synth.normal(3, "g3/x/" + name, (257, kw["in_channels_xyz"] + kw["in_channels_dir"]))
    fk = dict(D=kw["D"], W=kw["W"], in_channels_xyz=kw["in_channels_xyz"], in_channels_dir=kw["in_channels_dir"],
              raw_feat=kw["raw_feat"])
    assert rel_err(orc.nerf_forward(p, x, **fk), g[name]) < TOL
    assert rel_err(orc.nerf_forward(p, x[:, :kw["in_channels_xyz"]], sigma_only=True, **fk), g[name + "_sigma"]) < TOL


@pytest.mark.parametrize("B", [25, 36])
def test_g4_skinning(B):
    g = golden("g4_skinning")
    N, S = 12, 9
    bones = synth.make_models(4, B=B, with_skin=False, perturb_bones=True)["bones_rst"]
    rts = synth.frame_dual_quats(4, f"g4/rts{B}", N, B)
    xyz = np.float32(0.2) * synth.normal(4, f"g4/xyz{B}", (N, S, 3))
    dskin = synth.normal(4, f"g4/dskin{B}", (N, S, B))
    aux = np.asarray([0.3, 10], np.float32)
    bd = orc.bone_transform(bones, rts)
    gb = g[f"bone_transform_{B}"]
    # orientation sign is a convention of the absent pytorch3d (consumed only through q -> R(q), invariant to -q)
    assert rel_err(bd[..., :3], gb[..., :3]) < TOL and rel_err(bd[..., 7:], gb[..., 7:]) < TOL
    assert rel_err(orc.quaternion_to_matrix(bd[..., 3:7]), orc.quaternion_to_matrix(gb[..., 3:7])) < 5e-6
    # logits are O(1e3 * dist^2): fp32 round-off of ~1e-4 in a logit is ~1e-5 relative in the softmax
    assert rel_err(orc.skinning(bd, xyz, dskin, aux), g[f"skin_ray_dskin_{B}"]) < 5e-5
    assert rel_err(orc.skinning(bd, xyz, None, aux), g[f"skin_ray_{B}"]) < 5e-5
    assert rel_err(orc.skinning(bones, xyz, dskin, aux), g[f"skin_rest_dskin_{B}"]) < 5e-5
    skin = g[f"skin_ray_dskin_{B}"]
    assert rel_err(orc.dqs_blend_skinning(rts.reshape(N, B, 8), skin, xyz), g[f"dqs_{B}"]) < TOL
    assert rel_err(orc.neu_dbs(bones, rts, skin, xyz, backward=True), g[f"neu_dbs_bw_{B}"]) < TOL
    assert rel_err(orc.neu_dbs(bones, rts, skin, xyz, backward=False), g[f"neu_dbs_fw_{B}"]) < TOL
    # one-hot weights reproduce the rigid transform R p + t of that bone, and dq_inverse undoes it
    onehot = np.zeros((N, S, B), np.float32); onehot[..., 3] = 1
    fw = orc.neu_dbs(bones, rts, onehot, xyz, backward=False)
    assert rel_err(orc.neu_dbs(bones, rts, onehot, fw, backward=True), xyz) < 1e-5


def test_g5_composite():
    g = golden("g5_composite")
    N, S = 9, 12
    scene = oracle_scene(5, 0)
    rays = synth.make_rays(5, N, 0)
    z, xyz = g["z"], g["xyz"]
    d_emb = orc.embedding(rays["rays_d"], 4, 10.0)  # rays_d is already unit here
    fn = lambda x, sigma_only=False: orc.nerf_forward(scene.coarse, x, in_channels_dir=91)
    out = orc.evaluate_mlp(fn, xyz, embed_fn=lambda x: orc.embedding(x, 10, 10.0),
                           dir_embedded=np.broadcast_to(d_emb[:, None], (N, S, 27)), code=rays["env_code"], chunk=4096)
    names = ("rgb", "feat", "depth", "weights", "vis", "sil")
    o1 = orc.composite(out[..., :3], out[..., 3], np.zeros_like(out[..., :3]), z, rays["rays_d"], 0.1,
                       noise=g["noise_randn"] * np.float32(0.5))
    vis_pred = synth.uniform(5, "g5/vis", (N, S))
    oob = (np.abs(xyz) > np.asarray([0.12, 0.12, 0.25], np.float32)).sum(-1) > 0
    o2 = orc.composite(out[..., :3], out[..., 3], np.zeros_like(out[..., :3]), z, rays["rays_d"], 0.1,
                       oob=oob, vis_pred=vis_pred)
    for tag, o in (("noise", o1), ("mask", o2)):
        for n, v in zip(names, o):
            assert rel_err(v, g[f"{tag}_{n}"]) < 2e-5, (tag, n)
    assert oob.any() and (~oob).any()


def test_g6_sample_pdf():
    g = golden("g6_sample_pdf")
    N, S = 11, 14
    bins = np.sort(synth.uniform(6, "g6/bins", (N, S + 1)), -1).astype(np.float32)
    w = synth.uniform(6, "g6/w", (N, S)).astype(np.float32)
    w[2] = 0
    w[4, 3:9] = 0
    u = synth.uniform(6, "g6/u", (N, 20))
    assert rel_err(orc.sample_pdf(bins, w, 20), g["det"]) < 1e-5
    assert rel_err(orc.sample_pdf(bins, w, 20, u=u), g["rnd"]) < 1e-5


def run_oracle_case(name, dtype=np.float32, round_fn=None, seed=7, N=64, rays_per_frame=16):
    case = dict(E2E_CASES[name])
    g = golden("g7_" + name)
    B = case["B"]
    S = case.get("S", 16)
    scene = oracle_scene(seed, B, with_skin=case.get("with_skin", True), with_feat=case.get("with_feat", False),
                         with_vis=case.get("with_vis", False), alpha=case.get("alpha", 10.0),
                         perturb_bones=case.get("perturb_bones", False), dtype=dtype, with_dis=case.get("with_dis", False))
    rays = cast(synth.make_rays(seed, N, B, rays_per_frame=rays_per_frame), dtype)
    rnd = e2e_random_inputs(g, case)
    noise_std = {"perturb": 0.3, "fine_perturb_symm": 0.2}.get(name, 0.0)
    kw = dict(N_samples=S, use_disp=case.get("use_disp", False), perturb=case.get("perturb", 0),
              use_fine=case.get("use_fine", False), render_vis=case.get("render_vis", False),
              obj_bound=case.get("obj_bound"), perturb_rand=rnd.get("perturb_rand"), pdf_u=rnd.get("pdf_u"),
              symm_mask=rnd.get("symm_mask"), symm_mask_pre=rnd.get("symm_mask_pre"),
              noise=(rnd["noise_raw"] * np.float32(noise_std)).astype(dtype),
              noise_pre=(rnd["noise_pre_raw"] * np.float32(noise_std)).astype(dtype) if "noise_pre_raw" in rnd else None,
              round_fn=round_fn, rgb_filter_scale=1.3 if case.get("rgb_filter") else 0.0)    # make_opts: scale_rgb = 1.3
    return orc.render_rays(scene, rays, **kw), g


@pytest.mark.parametrize("name", list(E2E_CASES))
def test_g7_end_to_end(name):
    res, g = run_oracle_case(name)
    keys = [k for k in g if not k.startswith("rng")]
    assert "img_coarse" in keys
    for k in keys:
        assert res[k].shape == g[k].shape, k
        # fp32-vs-fp32 through 8 layers and a 1/beta = 10x density gain: 1e-4 is the north-star bar
        assert rel_err(res[k], g[k]) < 1e-4, (k, rel_err(res[k], g[k]))


def test_g7_float64_truth_brackets_reference():
    """The float64 oracle is a tighter truth: the fp32 reference sits within fp32 round-off of it."""
    res, g = run_oracle_case("bones_skin", dtype=np.float64)
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "xyz_canonical_vis", "frame_cyc_dis"):
        assert rel_err(res[k], g[k]) < 1e-4, k


def test_g8_cfg1_checksum():
    """BASELINE config 1 (4096 x 64, B=25) at full size: pins the benchmark's input + the oracle at scale."""
    g = golden("g8_cfg1")
    scene = oracle_scene(0, 25)
    rays = synth.make_rays(0, 4096, 25, rays_per_frame=256)
    res = orc.render_rays(scene, rays, N_samples=64, noise=None)
    idx = g["ray_index"]
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "xyz_camera_vis", "xyz_canonical_vis", "frame_cyc_dis"):
        assert rel_err(res[k][idx], g[k + "_rays"]) < 1e-4, k
        assert abs(res[k].astype(np.float64).mean() - g[k + "_mean"]) < 1e-4 * max(abs(g[k + "_mean"]), 1e-3), k
        assert abs(np.abs(res[k]).max() - g[k + "_absmax"]) < 1e-4 * g[k + "_absmax"], k


def _fn(p, kw, raw):
    return lambda x, sigma_only=False: orc.nerf_forward(p, x, D=kw["D"], W=kw["W"], in_channels_xyz=kw["in_channels_xyz"],
                                                        in_channels_dir=kw["in_channels_dir"], raw_feat=raw, sigma_only=sigma_only)


def test_g18_evaluate_mlp_wrapper():
    """orc.evaluate_mlp (geom_utils.py:19-57) in every calling form the reference uses, against the reference's outputs."""
    g = golden("g18_evaluate_mlp")
    N, S = 7, 9
    i = synth.evaluate_mlp_inputs(18, N, S)
    mp = synth.make_models(18, B=25, with_skin=True, with_feat=True, with_vis=True)
    mp_app = synth.make_models(18, B=0, with_app=True)
    emb = lambda x: orc.embedding(x, 10, 10.0)
    dir_e = np.repeat(orc.embedding(i["dirs"], 4, 10.0)[:, None], S, 1)
    kc = dict(NERF_SHAPES["coarse"])
    ka = dict(kc, in_channels_dir=27 + 64 + 128)
    got = {
        "coarse": orc.evaluate_mlp(_fn(mp["coarse"], kc, False), i["xyz"], embed_fn=emb, dir_embedded=dir_e, code=i["env"], chunk=3),
        "coarse_app": orc.evaluate_mlp(_fn(mp_app["coarse"], ka, False), i["xyz"], embed_fn=emb, dir_embedded=dir_e,
                                       code=i["env"][:, None], appearance_code=i["app"]),
        "coarse_sigma": orc.evaluate_mlp(_fn(mp["coarse"], kc, False), i["xyz"], embed_fn=emb, sigma_only=True),
        "skin_ray": orc.evaluate_mlp(_fn(mp["nerf_skin"], NERF_SHAPES["skin"], True), i["xyz"], embed_fn=emb,
                                     code=i["tcode"][:, None], chunk=2),
        "skin_rest": orc.evaluate_mlp(_fn(mp["nerf_skin"], NERF_SHAPES["skin"], True), i["xyz"], embed_fn=emb, code=i["rest"]),
        "skin_embedded": orc.evaluate_mlp(_fn(mp["nerf_skin"], NERF_SHAPES["skin"], True), emb(i["xyz"]), code=i["tcode"]),
        "feat": orc.evaluate_mlp(_fn(mp["nerf_feat"], NERF_SHAPES["feat"], True), i["xyz"], embed_fn=emb),
        "vis_embedded": orc.evaluate_mlp(_fn(mp["nerf_vis"], NERF_SHAPES["vis"], True), emb(i["xyz"]), chunk=5),
    }
    for k, v in got.items():
        assert v.shape == g[k].shape, k
        assert rel_err(v, g[k]) < 5e-6, (k, rel_err(v, g[k]))


def test_g19_uncertainty_head_and_appearance_code():
    """nerf_unc -> unc_pred (rendering.py:501-516) and the appearance-code columns of the colour branch (:369-372)."""
    g = golden("g19_unc_app_eval")
    N, S, B = 48, 12, 25
    mp = unc_scene_params(19)
    scene = orc.Scene(mp["coarse"], bones_rst=mp["bones_rst"], skin_aux=mp["skin_aux"], nerf_skin=mp["nerf_skin"],
                      rest_pose_code=mp["rest_pose_code"], alpha_xyz=10.0, alpha_dir=10.0, nerf_unc=mp["nerf_unc"])
    rays = synth.make_rays(19, N, B, rays_per_frame=8, with_app=True)
    rays.update(synth.make_unc_rays(19, N, 8))
    res = orc.render_rays(scene, rays, N_samples=S)
    for k in ("img_coarse", "sil_coarse", "depth_rnd", "unc_pred", "frame_cyc_dis"):
        assert res[k].shape == g[k].shape, k
        assert rel_err(res[k], g[k]) < 1e-4, (k, rel_err(res[k], g[k]))
        assert elem_err(res[k], g[k]) < 1, (k, elem_err(res[k], g[k]))


def test_g20_checkpoint_parameters_render_like_the_reference():
    """The oracle, fed the checkpoint fixture's tensors under the reference's key names and the per-frame codes the
    reference's FrameCode / DQ_RTHead produced from them, renders what the reference rendered."""
    g = golden("g20_checkpoint")
    sd = checkpoint_states(g)
    sub = lambda pre: {k[len(pre) + 1:]: v for k, v in sd.items() if k.startswith(pre + ".")}
    scene = orc.Scene(sub("nerf_coarse"), bones_rst=sd["bones"], skin_aux=sd["skin_aux"], nerf_skin=sub("nerf_skin"),
                      rest_pose_code=sd["rest_pose_code.weight"], nerf_vis=sub("nerf_vis"), nerf_feat=sub("nerf_feat"),
                      alpha_xyz=10.0, alpha_dir=10.0, nerf_unc=sub("nerf_unc"))
    N, S, F = 32, 12, 8
    rays = synth.make_rays(20, N, 0, rays_per_frame=N // F)
    rep = lambda a: np.repeat(a, N // F, 0)
    rays["bone_rts"], rays["time_embedded"], rays["env_code"] = rep(g["bone_rts"]), rep(g["time_embedded"]), rep(g["env_code"])
    rays.update(synth.make_unc_rays(20, N, N // F))
    rays["vid_code"] = g["vid_code"]
    res = orc.render_rays(scene, rays, N_samples=S, render_vis=True, obj_bound=[0.3, 0.3, 0.3])
    for k in ("img_coarse", "sil_coarse", "depth_rnd", "xyz_canonical_vis", "frame_cyc_dis", "vis_pred", "unc_pred"):
        assert rel_err(res[k], g["render_" + k]) < 1e-4, (k, rel_err(res[k], g["render_" + k]))


@pytest.mark.parametrize("case", ["small_eval", "small_train", "large_eval", "large_train"])
def test_g23_s3im_restatements_match_reference(case):
    """The S3IM term (loss_utils.py:575-702) restated in numpy and in torch against the reference's own value, on the rendered
    colours the reference returned (already masked by sil_at_samp -- a 0/1 mask, so masking again changes nothing) with the
    permutations it drew."""
    import torch
    from oracle import torch_ref as tr
    from moda_amd import synth
    g = golden("g23_s3im_" + case)
    N = g["img_coarse"].shape[0]
    mask = synth.make_corresp_rays(23, N, 25, rays_per_frame=4)["sil_at_samp"]
    assert set(np.unique(mask)) <= {0.0, 1.0}
    got = orc.s3im_loss(g["img_coarse"], g["img_at_samp"], mask, g["perms"])
    assert abs(got - float(g["s3im_loss"])) < 2e-6 * abs(float(g["s3im_loss"])) + 1e-7, (got, float(g["s3im_loss"]))
    t = tr.s3im_loss(torch.from_numpy(g["img_coarse"]), torch.from_numpy(g["img_at_samp"]), torch.from_numpy(mask),
                     torch.from_numpy(g["perms"]))
    assert abs(float(t) - float(g["s3im_loss"])) < 1e-5
    # the observed image comes back masked in place (loss_utils.py:666)
    assert np.array_equal(g["img_at_samp"], synth.make_corresp_rays(23, N, 25, rays_per_frame=4)["img_at_samp"] * mask)


def test_g25_cfg3_36_bones_symmetric_shape_combined():
    """BASELINE configs[2] in one fixture (round 4): 36 perturbed bones + the symmetric-shape flip (recorded mask), 32 samples."""
    g = golden("g25_cfg3_eval")
    scene = oracle_scene(25, 36, perturb_bones=True)
    rays = synth.make_rays(25, 64, 36, rays_per_frame=16)
    res = orc.render_rays(scene, rays, N_samples=32, symm_mask=g["rng0_rand_like"] < 0.5, noise=None)
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "xyz_camera_vis", "xyz_canonical_vis", "frame_cyc_dis"):
        assert res[k].shape == g[k].shape, k
        assert rel_err(res[k], g[k]) < 1e-4, (k, rel_err(res[k], g[k]))


def test_g24_cfg5_hierarchical_with_feature_net_combined():
    """BASELINE configs[4] in one fixture (round 4): the rendered outputs of the hierarchical 64 + 64 pass with the CSE feature net
    present (the matching / reprojection heads of the fixture are held by the torch restatement's counterparts, G11)."""
    g = golden("g24_cfg5_eval")
    scene = oracle_scene(24, 25, with_feat=True, with_vis=True, perturb_bones=True)
    rays = synth.make_rays(24, 32, 25, rays_per_frame=8)
    res = orc.render_rays(scene, rays, N_samples=128, use_fine=True, noise=None, noise_pre=None)     # 64 coarse + 64 importance (rendering.py:50)
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "xyz_camera_vis", "xyz_canonical_vis", "frame_cyc_dis"):
        assert res[k].shape == g[k].shape, (k, res[k].shape, g[k].shape)
        assert rel_err(res[k], g[k]) < 1e-4, (k, rel_err(res[k], g[k]))
