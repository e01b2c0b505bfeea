"""GPU (-m gpu): backward kernels / autograd Functions against torch autograd on the CPU restatement
(oracle/torch_ref.py) and against golden gradients produced by the reference's own autograd (g9)."""
import numpy as np
import pytest
import torch

from helpers import assert_gradients_within_f64_truth, golden, rel_err

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import moda_amd
    from moda_amd import synth, autograd as A
    from oracle import torch_ref as tr
    from gpu_helpers import T, DEV, make_models, make_opts, rays_to_gpu
    from test_torch_ref import g9_loss, check_grad, GRAD_LEAVES

TC = torch.from_numpy


def np_(t):
    return t.detach().cpu().numpy()


@pytest.fixture(params=["fp32", "bf16x6"])
def parity_precision(request):
    """The two parity-grade precisions of the training GEMMs: exact fp32 MFMA, and split-bf16 with three bf16 per operand
    (hi + mid + lo = the fp32 value, six bf16 MFMAs per product -- MODA_GEMM_BF16X6).  Tests that take this fixture hold both
    to the SAME bars."""
    moda_amd.set_train_precision(request.param)
    yield request.param
    moda_amd.set_train_precision("fp32")


@pytest.mark.parametrize("M,K,O,act", [(257, 63, 256, 1), (1000, 319, 256, 1), (37, 256, 1, 0), (4099, 128, 3, 2), (5, 347, 128, 1),
                                       (1000, 256, 256, 1), (4099, 128, 64, 1), (300, 64, 256, 0), (16500, 64, 64, 0), (130, 256, 100, 1)])
def test_linear_fn_forward_backward(M, K, O, act, parity_precision):
    x = synth.normal(31, "lin/x", (M, K)); W = synth.normal(31, "lin/w", (O, K)) * np.float32(0.1); b = synth.normal(31, "lin/b", (O,))
    g = synth.normal(31, "lin/g", (M, O))
    xc, Wc, bc = (TC(a).requires_grad_(True) for a in (x, W, b))
    z = xc @ Wc.T + bc
    yc = torch.relu(z) if act == 1 else (torch.sigmoid(z) if act == 2 else z)
    (yc * TC(g)).sum().backward()
    xg, Wg, bg = (T(a).requires_grad_(True) for a in (x, W, b))
    yg = A.LinearFn.apply(xg, Wg, bg, act)
    (yg * T(g)).sum().backward()
    assert rel_err(np_(yg), yc.detach().numpy()) < 2e-6
    assert rel_err(np_(xg.grad), xc.grad.numpy()) < 5e-6
    assert rel_err(np_(Wg.grad), Wc.grad.numpy()) < 2e-5      # split-K + atomics: different summation order
    assert rel_err(np_(bg.grad), bc.grad.numpy()) < 2e-5


@pytest.mark.parametrize("mode,bar", [("bf16x3", 2e-5), ("bf16x6", 1e-6)])   # x3: 2^-16 per product, no averaging at K = 2
@pytest.mark.parametrize("M,K,N", [(1000, 256, 256), (129, 64, 64), (40000, 128, 128), (5000, 256, 64), (777, 64, 128), (260, 128, 100),
                                   (333, 72, 36), (100, 8, 4), (2, 40, 200)])
def test_gemm_split_bf16_forms_against_float64(M, K, N, mode, bar):
    """The three forms gemm_x3.hip takes in the split-bf16 modes (MODA_GEMM_BF16X3: two bf16 per operand, 2^-17; MODA_GEMM_BF16X6:
    three, the fp32 value exactly -- bar = that of an fp32 GEMM) (forward: both operands k-fast, bias + ReLU; dX: n-fast
    weight, ReLU mask, C +=; dW: m-fast A, n-fast B, atomics over k) against float64, on ragged M; and an operand the router
    refuses (not 16-byte aligned) landing on the generic kernel's split-bf16 path."""
    x = synth.normal(61, "x3/x", (M, K)); w = np.float32(0.1) * synth.normal(61, "x3/w", (N, K)); b = synth.normal(61, "x3/b", (N,))
    dz = synth.normal(61, "x3/dz", (M, N)); c0 = synth.normal(61, "x3/c", (M, K))
    xg, wg, bg, dzg, cg = (T(a) for a in (x, w, b, dz, c0))
    moda_amd.set_train_precision(mode)
    try:
        y = A.gemm(xg, wg.t(), bias=bg, act=1)                                   # forward
        dx = A.gemm(dzg, wg, mask_src=xg, out=cg.clone(), accumulate=2)          # dX, masked, added to a running gradient
        dW = A.gemm(dzg.t(), xg, out=torch.zeros(N, K, device=DEV), accumulate=True, split_k=8)
        # the generic kernel's split-bf16 path: a column slice with an odd leading offset is not 16-byte aligned
        xo = T(np.concatenate([np.zeros((M, 1), np.float32), x], 1))[:, 1:]
        y2 = A.gemm(xo, wg.t(), bias=bg, act=1)
    finally:
        moda_amd.set_train_precision("fp32")
    x64, w64, dz64 = x.astype(np.float64), w.astype(np.float64), dz.astype(np.float64)
    y_ref = np.maximum(x64 @ w64.T + b, 0)
    assert rel_err(np_(y), y_ref) < bar, rel_err(np_(y), y_ref)
    assert rel_err(np_(y2), y_ref) < bar
    dx_ref = (c0 + dz64 @ w64) * (x > 0)                                          # moda_gemm_desc: C +=, then the mask
    assert rel_err(np_(dx), dx_ref) < bar, rel_err(np_(dx), dx_ref)
    dW_ref = dz64.T @ x64
    assert rel_err(np_(dW), dW_ref) < bar, rel_err(np_(dW), dW_ref)
    assert (np_(dx)[x <= 0] == 0).all()


def test_gemm_strided_views():
    a = synth.normal(32, "g/a", (300, 200)); b = synth.normal(32, "g/b", (150, 200))
    A_, B_ = T(a), T(b)
    got = A.gemm(A_[:, 10:170], B_[:, 20:180].t())            # non-unit leading strides, transposed view
    assert rel_err(np_(got), a[:, 10:170] @ b[:, 20:180].T) < 2e-6
    got = A.gemm(A_.t()[:64], A_[:, :77])                      # A transposed view (unit stride on m)
    assert rel_err(np_(got), a.T[:64] @ a[:, :77]) < 2e-6


def test_embed_fn_grad():
    x = np.float32(0.4) * synth.normal(33, "e/x", (129, 3)); g = synth.normal(33, "e/g", (129, 63))
    for normalize, nf, alpha in ((False, 10, 10.0), (False, 10, 6.5), (True, 4, 4.0)):
        xc = TC(x).requires_grad_(True)
        u = xc / xc.norm(dim=-1, keepdim=True) if normalize else xc
        ec = tr.embedding(u, nf, alpha)
        gg = g[:, :ec.shape[1]]
        (ec * TC(gg)).sum().backward()
        xg = T(x).requires_grad_(True)
        eg = moda_amd.Embedding(3, nf, alpha=alpha)(xg, normalize=normalize)
        (eg * T(gg)).sum().backward()
        assert rel_err(np_(eg), ec.detach().numpy()) < 2e-6
        assert rel_err(np_(xg.grad), xc.grad.numpy()) < 2e-5, (normalize, nf)


def test_composite_fn_grad():
    N, S, F = 21, 150, 16      # S > 64: the reverse scan carries its suffix across blocks
    rgbs = synth.uniform(34, "c/rgb", (N, S, 3)); sig = np.float32(0.05) * synth.normal(34, "c/sig", (N, S))
    feat = synth.normal(34, "c/feat", (N, S, F))
    z = np.sort(np.float32(0.1) + np.float32(0.4) * synth.uniform(34, "c/z", (N, S)), -1).astype(np.float32)
    rd = synth.normal(34, "c/rd", (N, 3)); cyc = synth.uniform(34, "c/cyc", (N, S)); noise = np.float32(0.02) * synth.normal(34, "c/n", (N, S))
    gs = {k: synth.normal(34, "c/g" + k, s) for k, s in (("rgb", (N, 3)), ("feat", (N, F)), ("depth", (N,)), ("sil", (N,)), ("cyc", (N,)))}
    leaf = lambda a: TC(a).requires_grad_(True)
    c_rgbs, c_sig, c_feat, c_rd, c_cyc, c_beta = leaf(rgbs), leaf(sig), leaf(feat), leaf(rd), leaf(cyc), leaf(np.asarray([0.1], np.float32))
    rgb, fo, depth, w, _, sil = tr.composite(c_rgbs, c_sig, c_feat, TC(z), c_rd, c_beta, TC(noise))
    co = (c_cyc * w.detach()).sum(-1)
    loss = (rgb * TC(gs["rgb"])).sum() + (fo * TC(gs["feat"])).sum() + (depth * TC(gs["depth"])).sum() + (sil * TC(gs["sil"])).sum() \
        + (co * TC(gs["cyc"])).sum()
    loss.backward()
    g_rs = T(np.concatenate([rgbs, sig[..., None]], -1)).requires_grad_(True)
    g_feat, g_rd, g_cyc, g_beta = (T(a).requires_grad_(True) for a in (feat, rd, cyc, np.asarray([0.1], np.float32)))
    o = A.CompositeFn.apply(g_rs, g_feat, T(z), g_rd, g_beta, T(noise), None, None, None, g_cyc)
    lg = (o[0] * T(gs["rgb"])).sum() + (o[1] * T(gs["feat"])).sum() + (o[2] * T(gs["depth"])).sum() + (o[3] * T(gs["sil"])).sum() \
        + (o[7] * T(gs["cyc"])).sum()
    lg.backward()
    assert abs(float(lg) - float(loss)) < 1e-4 * abs(float(loss))
    ref_rs = np.concatenate([c_rgbs.grad.numpy(), c_sig.grad.numpy()[..., None]], -1)
    assert rel_err(np_(g_rs.grad), ref_rs) < 1e-4
    assert rel_err(np_(g_feat.grad), c_feat.grad.numpy()) < 1e-5
    assert rel_err(np_(g_rd.grad), c_rd.grad.numpy()) < 1e-4
    assert rel_err(np_(g_cyc.grad), c_cyc.grad.numpy()) < 1e-5
    assert rel_err(np_(g_beta.grad), c_beta.grad.numpy()) < 1e-4


@pytest.mark.parametrize("per_ray,B", [(True, 25), (False, 36)])
def test_warp_fn_grad(per_ray, B):
    N, S = 9, 70
    bones = synth.make_models(35, B=B, with_skin=False, perturb_bones=True)["bones_rst"]
    rts = synth.frame_dual_quats(35, "w/rts", N, B).reshape(N, B, 8)
    xyz = np.float32(0.15) * synth.normal(35, "w/xyz", (N, S, 3)); dskin = synth.normal(35, "w/ds", (N, S, B))
    ref = np.float32(0.15) * synth.normal(35, "w/ref", (N, S, 3))
    g_out = synth.normal(35, "w/go", (N, S, 3)); g_cyc = synth.normal(35, "w/gc", (N, S))
    aux = np.asarray([0.2, 10], np.float32)

    def run(conv, dev_fn):
        b_, r_, x_, d_, a_, f_ = (conv(a).requires_grad_(True) for a in (bones, rts, xyz, dskin, aux, ref))
        return dev_fn(b_, r_, x_, d_, a_, f_) + ((b_, r_, x_, d_, a_, f_),)

    def cpu_fn(b_, r_, x_, d_, a_, f_):
        bset = tr.bone_transform(b_, r_) if per_ray else b_
        skin = tr.skinning(bset, x_, d_, a_)
        out = tr.dqs(tr.dq_inverse(r_) if per_ray else r_, skin, x_)
        return out, (f_ - out).norm(dim=-1)

    def gpu_fn(b_, r_, x_, d_, a_, f_):
        prep = A.bone_prep(A.bone_transform(b_, r_)) if per_ray else A.bone_prep(b_[None])
        out, cyc, _ = A.WarpFn.apply(prep, A.dq_inverse(r_) if per_ray else r_, x_, d_, a_, f_)
        return out, cyc

    oc, cc, lc = run(TC, cpu_fn)
    ((oc * TC(g_out)).sum() + (cc * TC(g_cyc)).sum()).backward()
    og, cg, lg = run(T, gpu_fn)
    ((og * T(g_out)).sum() + (cg * T(g_cyc)).sum()).backward()
    assert rel_err(np_(og), oc.detach().numpy()) < 1e-5 and rel_err(np_(cg), cc.detach().numpy()) < 1e-5
    for name, a, b in zip(("bones", "rts", "xyz", "dskin", "aux", "ref"), lg, lc):
        assert rel_err(np_(a.grad), b.grad.numpy()) < 2e-4, (name, rel_err(np_(a.grad), b.grad.numpy()))


@pytest.mark.parametrize("case,B,with_skin", [("nobones", 0, False), ("bones_noskin", 25, False), ("bones_skin", 25, True)])
def test_render_rays_gradients_match_reference_autograd(case, B, with_skin, parity_precision):
    """End to end: d(loss)/d(parameters, ray inputs) through moda_amd.render_rays vs the REFERENCE's autograd (g9)."""
    g = golden("g9_grad_" + case)
    models, emb = make_models(9, B, with_skin=with_skin, perturb_bones=True)
    for m in models.values():
        if isinstance(m, torch.nn.Module):
            m.train()
    if B > 0:
        models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
        models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
    rays = rays_to_gpu(synth.make_rays(9, 48, B, rays_per_frame=8))
    for k in GRAD_LEAVES:
        if k in rays:
            rays[k].requires_grad_(True)
    res = moda_amd.render_rays(models, emb, rays, N_samples=12, noise_std=0.0, opts=make_opts(), img_size=512)
    loss = 0
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "frame_cyc_dis"):
        if k in res:
            loss = loss + (T(synth.normal(9, "g9/c/" + k, tuple(res[k].shape))) * res[k]).sum()
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    # fp32 gradients vs the reference's autograd, relative L2 error per tensor.  The fixture has only 576 samples, so
    # ONE sample whose ReLU pre-activation is ~0 and switches between two correct fp32 evaluations moves a weight
    # gradient by ~1/576 = 2e-3 of its norm; the per-Function tests above are the tight (1e-5..1e-4) checks.
    tol = 2e-3 if B == 0 else 1e-2
    errs = {}
    for k in GRAD_LEAVES:
        if "d_" + k in g:
            errs[k] = rel_err(np_(rays[k].grad), g["d_" + k])
    print(case, {k: f"{v:.1e}" for k, v in errs.items()})
    for k in GRAD_LEAVES:
        if "d_" + k in g:
            check_grad("d_" + k, np_(rays[k].grad), g, tol, l2=True)
    for pn, p in models["coarse"].named_parameters():
        if p.grad is not None:
            check_grad("d_coarse." + pn, np_(p.grad), g, tol, l2=True)
    if B > 0:
        check_grad("d_bones_rst", np_(models["bones_rst"].grad), g, tol, l2=True)
        check_grad("d_skin_aux", np_(models["skin_aux"].grad), g, tol, l2=True)
    if with_skin:
        check_grad("d_rest_pose_code", np_(models["rest_pose_code"].weight.grad), g, tol, l2=True)
        for pn, p in models["nerf_skin"].named_parameters():
            if p.grad is not None and ("d_nerf_skin." + pn in g or "d_nerf_skin." + pn + "__corner" in g):
                check_grad("d_nerf_skin." + pn, np_(p.grad), g, tol, l2=True)
    # the float64 truth (the reference run in float64, g9_grad_*_f64.npz): no further from it than the reference's own fp32
    # autograd allows, and inside the absolute bar -- helpers.assert_gradients_within_f64_truth
    got = {"d_" + k: np_(rays[k].grad) for k in GRAD_LEAVES if k in rays and rays[k].grad is not None}
    for mn in ("coarse", "nerf_skin"):
        if mn in models:
            got.update({f"d_{mn}.{pn}": np_(p.grad) for pn, p in models[mn].named_parameters() if p.grad is not None})
    if B > 0:
        got["d_bones_rst"], got["d_skin_aux"] = np_(models["bones_rst"].grad), np_(models["skin_aux"].grad)
    if with_skin:
        got["d_rest_pose_code"] = np_(models["rest_pose_code"].weight.grad)
    assert_gradients_within_f64_truth("g9_grad_" + case, got, parity_precision, 48, 12)


def test_api_functions_under_autograd():
    """The function-level API (skinning, bone_transform, dual_quat) also differentiates."""
    from moda_amd import geom_utils as G, dual_quat as DQ
    B, N, S = 25, 6, 10
    bones = synth.make_models(36, B=B, with_skin=False, perturb_bones=True)["bones_rst"]
    rts = synth.frame_dual_quats(36, "api/rts", N, B)
    xyz = np.float32(0.15) * synth.normal(36, "api/xyz", (N, S, 3)); dskin = synth.normal(36, "api/ds", (N, S, B))
    aux = np.asarray([0.1, 10], np.float32); gs = synth.normal(36, "api/g", (N, S, B))
    cb, cr, cx, cd, ca = (TC(a).requires_grad_(True) for a in (bones, rts, xyz, dskin, aux))
    (tr.skinning(tr.bone_transform(cb, cr), cx, cd, ca) * TC(gs)).sum().backward()
    gb, gr, gx, gd, ga = (T(a).requires_grad_(True) for a in (bones, rts, xyz, dskin, aux))
    sk = G.skinning(G.bone_transform(gb, gr, True, is_vec=True), gx, gd, ga)
    (sk * T(gs)).sum().backward()
    for name, a, b in zip(("bones", "rts", "xyz", "dskin", "aux"), (gb, gr, gx, gd, ga), (cb, cr, cx, cd, ca)):
        assert rel_err(np_(a.grad), b.grad.numpy()) < 5e-4, name    # aux: a sum of O(1e3)-sized logit terms
    u = T(synth.frame_dual_quats(36, "api/u", 4, 5).reshape(20, 8)).requires_grad_(True)
    out = DQ.dq_mul(u, DQ.dq_inverse(u))
    out.sum().backward()
    assert torch.isfinite(u.grad).all() and rel_err(np_(out), np.tile(np.asarray([1, 0, 0, 0, 0, 0, 0, 0], np.float32), (20, 1))) < 1e-5


G10_KEYS = ("img_coarse", "sil_coarse", "flo_coarse", "flo_valid", "fdp_coarse", "fdp_valid", "img_loss_samp",
            "sil_loss_samp", "flo_loss_samp", "sil_at_samp_flo", "frame_cyc_dis")
G10_LOSS = ("flo_coarse", "fdp_coarse", "img_loss_samp", "sil_loss_samp", "flo_loss_samp", "frame_cyc_dis")


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_correspondence_and_loss_heads_match_reference(mode):
    """Paired-frame flow rendering + img/sil/flo loss terms (rendering.py:345-360, 439-499, 518-571) vs the reference:
    outputs in eval and train mode, gradients in train mode (tests/golden/g10_corresp_*.npz)."""
    from test_torch_ref import rel_l2
    g = golden("g10_corresp_" + mode)
    N, S, B = 48, 12, 25
    models, emb = make_models(10, B, with_skin=True, perturb_bones=True)
    if mode == "train":
        models["coarse"].train()
        models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
        models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
    rays = rays_to_gpu(synth.make_rays(10, N, B, rays_per_frame=8))
    rays.update(rays_to_gpu(synth.make_corresp_rays(10, N, B, rays_per_frame=8)))
    leaves = ("rays_o", "rays_d", "bone_rts", "bone_rts_target", "rtk_vec_target", "time_embedded")
    if mode == "train":
        for k in leaves:
            rays[k].requires_grad_(True)
    with (torch.enable_grad() if mode == "train" else torch.no_grad()):
        res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, img_size=512,
                                   opts=make_opts(dist_corresp=True, use_corresp=True))
    for k in G10_KEYS:
        got = np_(res[k].float())
        assert got.shape == g[k].shape, k
        assert rel_err(got, g[k]) < 2e-4, (mode, k, rel_err(got, g[k]))
    if mode == "train":
        loss = 0
        for k in G10_LOSS:
            loss = loss + (T(synth.normal(10, "g10/c/" + k, tuple(res[k].shape))) * res[k]).sum()
        assert abs(float(loss) - float(g["loss"])) < 2e-4 * abs(float(g["loss"]))
        loss.backward()
        got = {"d_" + k: rays[k].grad for k in leaves}
        got["d_bones_rst"] = models["bones_rst"].grad
        got["d_coarse.sigma.weight"] = models["coarse"].sigma.weight.grad
        got["d_coarse.xyz_encoding_1.0.weight"] = models["coarse"].xyz_encoding_1[0].weight.grad
        got["d_nerf_skin.rgb.0.weight"] = models["nerf_skin"].rgb[0].weight.grad
        for k, v in got.items():
            assert rel_l2(np_(v), g[k]) < 1e-2, (k, rel_l2(np_(v), g[k]))
        assert_gradients_within_f64_truth("g10_corresp_train", {k: np_(v) for k, v in got.items()}, "fp32", N, S)


def test_normalize_and_logsig_fns():
    x = synth.normal(41, "n/x", (133, 16)); x[7] = 0; g = synth.normal(41, "n/g", (133, 16))
    xc = TC(x).requires_grad_(True)
    yc = torch.nn.functional.normalize(xc, 2, -1)
    (yc * TC(g)).sum().backward()
    xg = T(x).requires_grad_(True)
    yg = A.NormalizeFn.apply(xg)
    (yg * T(g)).sum().backward()
    assert rel_err(np_(yg), yc.detach().numpy()) < 2e-6
    keep = np.arange(133) != 7     # the zero row: d/dx of x/max(|x|,eps) is g/eps on both sides, 1e12-sized
    assert rel_err(np_(xg.grad)[keep], xc.grad.numpy()[keep]) < 1e-5
    z = np.float32(3) * synth.normal(41, "l/z", (57, 19)); w = synth.uniform(41, "l/w", (57, 19))
    for sign, ww in ((1.0, w), (-1.0, None)):
        zc = TC(z).requires_grad_(True)
        lc = -(torch.nn.functional.logsigmoid(sign * zc) * (1 if ww is None else TC(ww))).sum() * 0.37
        lc.backward()
        zg = T(z).requires_grad_(True)
        lg = A.LogSigLossFn.apply(zg, None if ww is None else T(ww), sign, 0.37)
        (lg * 1.0).backward()
        assert abs(float(lg) - float(lc)) < 1e-5 * abs(float(lc))
        assert rel_err(np_(zg.grad), zc.grad.numpy()) < 1e-5


@pytest.mark.parametrize("use_ot", [True, False])
def test_feat_match_fn_grad(use_ot):
    """FeatMatchFn (Sinkhorn reverse sweep / softmax) against torch autograd on the plain restatement (float64)."""
    N, G = 70, 513
    f = synth.normal(42, "fm/f", (N, 16)); v = synth.normal(42, "fm/v", (G, 16)); q = np.float32(0.2) * synth.normal(42, "fm/q", (G, 3))
    gp = synth.normal(42, "fm/g", (N, 3))
    kap = np.asarray([1 / 0.03 if use_ot else 2.5], np.float32)

    def ref(fc, vc, kc):
        fn, vn = tr.normalize(fc), tr.normalize(vc)
        cost = fn @ vn.T
        if use_ot:
            K = torch.exp(-(1.0 - cost) / 0.03)
            a = torch.full((N, 1), 1.0 / N, dtype=fc.dtype)
            for _ in range(20):
                b = (1.0 / G) / (K.T @ a + 1e-8)
                a = (1.0 / N) / (K @ b + 1e-8)
            Tm = a * K * b.T
            prob = Tm / Tm.sum(1, keepdim=True)
        else:
            prob = (cost * kc).softmax(-1)
        return prob @ TC(q).to(fc.dtype)

    fc, vc, kc = (TC(a).double().requires_grad_(True) for a in (f, v, kap))
    pc = ref(fc, vc, kc)
    (pc * TC(gp).double()).sum().backward()
    fg, vg, kg = (T(a).requires_grad_(True) for a in (f, v, kap))
    pg = A.FeatMatchFn.apply(A.NormalizeFn.apply(fg), A.NormalizeFn.apply(vg), T(q), kg, use_ot)[0]
    (pg * T(gp)).sum().backward()
    assert rel_err(np_(pg), pc.detach().numpy()) < 2e-5
    assert rel_err(np_(fg.grad), fc.grad.numpy()) < 2e-4, rel_err(np_(fg.grad), fc.grad.numpy())
    assert rel_err(np_(vg.grad), vc.grad.numpy()) < 2e-4, rel_err(np_(vg.grad), vc.grad.numpy())
    if not use_ot:
        assert rel_err(np_(kg.grad), kc.grad.numpy()) < 2e-4


@pytest.mark.parametrize("use_ot", [True, False])
def test_feat_match_bf16_matrix_of_the_throughput_mode(use_ot):
    """In the bf16 training mode the matching matrix K (and its transpose) is held as bf16 (kmat_bf16 of the moda_match_*
    entries, ABI 5); vectors, sums and results stay fp32.  Against the same head in the fp32 mode, and against the float64
    restatement with K rounded to bf16 -- for the prediction the rounding is the only difference, so that comparison is tight.
    Sizes off the vector widths (G = 1003 columns: the 8-wide bf16 loads end in a scalar tail)."""
    N, G = 130, 1003
    f = synth.normal(43, "fmb/f", (N, 16)); v = synth.normal(43, "fmb/v", (G, 16)); q = np.float32(0.2) * synth.normal(43, "fmb/q", (G, 3))
    gp = synth.normal(43, "fmb/g", (N, 3))
    kap = np.asarray([1 / 0.03 if use_ot else 2.5], np.float32)

    def run(prec):
        fg, vg, kg = (T(a).requires_grad_(True) for a in (f, v, kap))
        moda_amd.set_train_precision(prec)
        try:
            pg = A.FeatMatchFn.apply(A.NormalizeFn.apply(fg), A.NormalizeFn.apply(vg), T(q), kg, use_ot)[0]
            (pg * T(gp)).sum().backward()
        finally:
            moda_amd.set_train_precision("fp32")
        return pg.detach(), fg.grad, vg.grad

    def ref_rounded():
        fc, vc, kc = (TC(a).double().requires_grad_(True) for a in (f, v, kap))
        fn, vn = tr.normalize(fc), tr.normalize(vc)
        cost = fn @ vn.T
        arg = (cost - 1.0) * kc
        # K as the kernels read it (rounded to bf16), with the derivative they use: d K / d arg = the rounded K
        K = torch.exp(arg).detach().float().bfloat16().double() * torch.exp(arg - arg.detach())
        if use_ot:
            a = torch.full((N, 1), 1.0 / N, dtype=torch.float64)
            for _ in range(20):
                b = (1.0 / G) / (K.T @ a + 1e-8)
                a = (1.0 / N) / (K @ b + 1e-8)
            Tm = a * K * b.T
            prob = Tm / Tm.sum(1, keepdim=True)
        else:
            prob = K / K.sum(1, keepdim=True)
        pc = prob @ TC(q).double()
        (pc * TC(gp).double()).sum().backward()
        return pc.detach(), fc.grad, vc.grad

    from helpers import rel_l2
    p32, f32g, v32g = run("fp32")
    p16, f16g, v16g = run("bf16")
    pr, frg, vrg = ref_rounded()
    for name, a, b, c in (("pred", p16, p32, pr), ("d_f", f16g, f32g, frg), ("d_v", v16g, v32g, vrg)):
        e_mode, e_ref = rel_l2(np_(a), np_(b)), rel_l2(np_(a), c.numpy())
        print(f"bf16 matching matrix ({'ot' if use_ot else 'softmax'}) {name}: vs fp32 mode {e_mode:.2e}, vs the rounded restatement {e_ref:.2e}")
        assert e_mode < 3e-2, (name, e_mode)
        # (the gradients also pass through the two bf16-operand GEMMs Dbar @ v and Dbar^T @ f of that mode)
        assert e_ref < (5e-4 if name == "pred" else 6e-3), (name, e_ref)


@pytest.mark.parametrize("N,G", [(2048, 8000), (512, 8000), (1024, 1000)])
def test_persistent_sinkhorn_equals_the_per_sweep_launches(N, G, monkeypatch):
    """moda_match_sinkhorn (round 5: the 40 forward and 38 backward sweeps of feat_match's Sinkhorn iterations as ONE persistent
    launch each way -- the matching matrix resident in LDS / registers, a flag-array grid barrier between the sweeps) against the
    chain of moda_match_sweep launches it replaces, in the bf16-matrix mode it serves: same prediction and gradients up to the
    sums' association (fp32), no time-out flag, and the launch count of the head drops by 76."""
    f = synth.normal(44, "ps/f", (N, 16)); v = synth.normal(44, "ps/v", (G, 16)); q = np.float32(0.2) * synth.normal(44, "ps/q", (G, 3))
    gp = synth.normal(44, "ps/g", (N, 3))
    kap = np.asarray([1 / 0.03], np.float32)

    def run(persist):
        monkeypatch.setattr(A, "SINKHORN_PERSIST", persist)
        calls = []
        orig = A.L.call

        def spy(name, *a):
            calls.append(name)
            return orig(name, *a)
        monkeypatch.setattr(A.L, "call", spy)
        fg, vg = (T(a).requires_grad_(True) for a in (f, v))
        moda_amd.set_train_precision("bf16")
        try:
            pg = A.FeatMatchFn.apply(A.NormalizeFn.apply(fg), A.NormalizeFn.apply(vg), T(q), T(kap), True)[0]
            (pg * T(gp)).sum().backward()
        finally:
            moda_amd.set_train_precision("fp32")
            monkeypatch.setattr(A.L, "call", orig)
        torch.cuda.synchronize()
        return pg.detach(), fg.grad, vg.grad, calls.count("moda_match_sweep")

    from helpers import rel_l2
    p0, f0, v0, n0 = run(False)
    p1, f1, v1, n1 = run(True)
    assert n0 == 78 and n1 == 0, (n0, n1)
    for name, a, b in (("pred", p1, p0), ("d_f", f1, f0), ("d_v", v1, v0)):
        e = rel_l2(np_(a), np_(b))
        print(f"persistent Sinkhorn N={N} G={G} {name}: rel-L2 vs the per-sweep chain {e:.2e}")
        assert torch.isfinite(a).all() and e < 2e-5, (name, e)


G11_BOUND = np.asarray([0.2, 0.2, 0.2], np.float32)
G11_KEYS = ("img_coarse", "sil_coarse", "pts_pred", "pts_exp", "feat_err", "proj_err", "vis_loss", "frnd_loss_samp",
            "flo_coarse", "img_loss_samp", "sil_loss_samp", "flo_loss_samp", "frame_cyc_dis")
G11_LOSS = ("pts_pred", "feat_err", "proj_err", "vis_loss", "frnd_loss_samp", "flo_coarse", "img_loss_samp",
            "sil_loss_samp", "frame_cyc_dis")


@pytest.mark.parametrize("mode,use_ot", [("eval_ot", True), ("train_ot", True), ("train_softmax", False)])
def test_full_training_configuration_heads_match_reference(mode, use_ot):
    """MoDA's default training configuration of inference_deform (use_embed, use_proj, use_corresp, dist_corresp,
    nerf_vis, use_ot): feature matching + keypoint reprojection + visibility loss + rendered-feature loss next to
    the flow / img / sil terms, outputs and gradients vs the reference (tests/golden/g11_heads_*.npz)."""
    from test_torch_ref import rel_l2
    g = golden("g11_heads_" + mode)
    train = mode.startswith("train")
    N, S, B = 48, 12, 25
    models, emb = make_models(11, B, with_skin=True, with_feat=True, with_vis=True, perturb_bones=True)
    if train:
        for m in models.values():
            if isinstance(m, torch.nn.Module):
                m.train()
        models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
        models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
    rays = rays_to_gpu(synth.make_rays(11, N, B, rays_per_frame=8))
    rays.update(rays_to_gpu(synth.make_corresp_rays(11, N, B, rays_per_frame=8)))
    rays.update(rays_to_gpu(synth.make_feat_rays(11, N, rays_per_frame=8)))
    leaves = ("rays_o", "rays_d", "bone_rts", "rtk_vec", "time_embedded")
    rng = None
    if train:
        for k in leaves:
            rays[k].requires_grad_(True)
        rng = {"feat_noise": T(g["rng_randn_like"]), "vis_neg_rand": TC(g["rng_rand"])}
    with (torch.enable_grad() if train else torch.no_grad()):
        res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, img_size=512, obj_bound=G11_BOUND,
                                   opts=make_opts(dist_corresp=True, use_corresp=True, use_ot=use_ot), rng=rng)
    for k in G11_KEYS:
        if k not in g:
            continue
        got = np_(res[k].float())
        assert got.shape == g[k].shape, (k, got.shape, g[k].shape)
        assert rel_err(got, g[k]) < 2e-4, (mode, k, rel_err(got, g[k]))
    if train:
        loss = 0
        for k in G11_LOSS:
            c = T(synth.normal(11, "g11/c/" + k, tuple(res[k].shape) or (1,))).reshape(res[k].shape)
            loss = loss + (c * res[k]).sum()
        assert abs(float(loss) - float(g["loss"])) < 2e-4 * abs(float(g["loss"]))
        loss.backward()
        got = {"d_" + k: rays[k].grad for k in leaves}
        got["d_bones_rst"] = models["bones_rst"].grad
        for mn, pn in (("nerf_feat", "rgb.0.weight"), ("nerf_feat", "xyz_encoding_1.0.weight"), ("nerf_vis", "rgb.0.weight"),
                       ("nerf_vis", "xyz_encoding_1.0.weight"), ("coarse", "sigma.weight"), ("nerf_skin", "rgb.0.weight")):
            got[f"d_{mn}.{pn}"] = dict(models[mn].named_parameters())[pn].grad
        if not use_ot:
            got["d_nerf_feat.beta"] = models["nerf_feat"].beta.grad
        for k, v in got.items():
            assert v is not None, k
            assert rel_l2(np_(v), g[k]) < 1e-2, (k, rel_l2(np_(v), g[k]))
        assert_gradients_within_f64_truth("g11_heads_" + mode, {k: np_(v) for k, v in got.items()}, "fp32", N, S)


def _relu_ambiguous(p, xyz, code, dirs, kw, sigma_only, tol=2e-6):
    """(R, S) bool: samples of a NeRF.forward (nerf.py:147-198) with a ReLU pre-activation within tol (relative to the layer's
    largest) of zero, evaluated in float64."""
    R, S = xyz.shape[:2]
    pc = {k: TC(v).double() for k, v in p.items()}
    cols = [tr.embedding(TC(xyz).double(), 10, 10.0)]
    for extra in (code, dirs):
        if extra is not None:
            cols.append(TC(extra).double()[:, None].expand(R, S, extra.shape[1]))
    x = torch.cat(cols, -1)
    in_xyz, in_dir = kw["in_channels_xyz"], kw["in_channels_dir"]
    amb = torch.zeros(R, S, dtype=torch.bool)

    def lin(t, n):
        z = t @ pc[n + ".weight"].T + pc[n + ".bias"]
        amb.logical_or_((z.abs() < tol * z.abs().max()).any(-1))
        return torch.relu(z)
    h = x[..., :in_xyz]
    for i in range(kw["D"]):
        if i == 4:
            h = torch.cat([x[..., :in_xyz], h], -1)
        h = lin(h, f"xyz_encoding_{i+1}.0")
    if not sigma_only:
        final = h @ pc["xyz_encoding_final.weight"].T + pc["xyz_encoding_final.bias"]
        lin(torch.cat([final, x[..., in_xyz:in_xyz + in_dir]], -1), "dir_encoding.0")
    return amb.numpy()


@pytest.mark.parametrize("name,kw,code_c,dir_c,sigma_only", [
    ("coarse", dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False), 0, 91, False),
    ("coarse_sigma", dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False), 0, 91, True),
    ("skin", dict(D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=25, raw_feat=True), 128, 0, False),
    ("feat", dict(D=5, W=128, in_channels_xyz=63, in_channels_dir=0, out_channels=16, raw_feat=True), 0, 0, False),
    ("vis", dict(D=5, W=64, in_channels_xyz=63, in_channels_dir=0, out_channels=1, raw_feat=True), 0, 0, False)])
def test_nerf_fn_whole_network_grad(name, kw, code_c, dir_c, sigma_only, parity_precision):
    """NerfFn (PE + all layers as one autograd node, per-ray inputs folded, skip layer read in place) against torch
    autograd on the plain restatement with the inputs concatenated per sample as the reference does."""
    from gpu_helpers import nerf_from_params
    R, S = 7, 37
    M = R * S
    pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
    p = synth.nerf_params(51, "nf/" + name, **pk)
    xyz = np.float32(0.3) * synth.normal(51, "nf/xyz", (R, S, 3))
    code = synth.normal(51, "nf/code", (R, code_c)) if code_c else None
    dirs = synth.normal(51, "nf/dir", (R, dir_c)) if dir_c else None
    n_out = 1 if sigma_only else kw["out_channels"] + (0 if kw["raw_feat"] else 1)
    gout = synth.normal(51, "nf/g", (R, S, n_out))
    # a ReLU whose pre-activation is zero to fp32 accuracy has no determined gradient: any two correct fp32 evaluations may
    # disagree on it, and one switched unit moves that sample's gradient by percents.  Samples holding such a unit (float64
    # |z| < 2e-6 of the layer's largest; this fixture has one, the skin net's layer 1 at 1e-10) get a zero output cotangent.
    gout[_relu_ambiguous(p, xyz, code, dirs, kw, sigma_only)] = 0
    # CPU restatement
    pc = {k: TC(v).requires_grad_(True) for k, v in p.items()}
    xc = TC(xyz).requires_grad_(True)
    cc = None if code is None else TC(code).requires_grad_(True)
    dc = None if dirs is None else TC(dirs).requires_grad_(True)
    cols = [tr.embedding(xc, 10, 10.0)]
    if cc is not None:
        cols.append(cc[:, None].expand(R, S, code_c))
    if dc is not None:
        cols.append(dc[:, None].expand(R, S, dir_c))
    yc = tr.nerf_forward(pc, torch.cat(cols, -1), kw["D"], kw["W"], kw["in_channels_xyz"], kw["in_channels_dir"],
                         raw_feat=kw["raw_feat"], sigma_only=sigma_only)
    (yc * TC(gout)).sum().backward()
    # GPU
    m = nerf_from_params(p, **kw).train()
    emb = moda_amd.Embedding(3, 10)
    xg = T(xyz).requires_grad_(True)
    cg = None if code is None else T(code).requires_grad_(True)
    dg = None if dirs is None else T(dirs).requires_grad_(True)
    yg = m.train_forward(xg, emb, code=cg, dir_src=dg, sigma_only=sigma_only)
    (yg * T(gout)).sum().backward()
    assert rel_err(np_(yg), yc.detach().numpy()) < 5e-6
    assert rel_err(np_(xg.grad), xc.grad.numpy()) < 1e-4, rel_err(np_(xg.grad), xc.grad.numpy())
    if cg is not None:
        assert rel_err(np_(cg.grad), cc.grad.numpy()) < 1e-4
    if dg is not None and not sigma_only:
        assert rel_err(np_(dg.grad), dc.grad.numpy()) < 1e-4
    for pn, pt in m.named_parameters():
        ref = pc[pn].grad
        if ref is None or float(ref.abs().max()) == 0:
            assert pt.grad is None or float(pt.grad.abs().max()) == 0, pn
            continue
        assert pt.grad is not None, pn
        assert rel_err(np_(pt.grad), ref.numpy()) < 1e-4, (pn, rel_err(np_(pt.grad), ref.numpy()))


@pytest.mark.parametrize("name,kw,code_c,dir_c", [
    ("coarse", dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False), 0, 91),
    ("skin", dict(D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=25, raw_feat=True), 128, 0),
    ("feat", dict(D=5, W=128, in_channels_xyz=63, in_channels_dir=0, out_channels=16, raw_feat=True), 0, 0),
    # more than 32 outputs (a 36-bone skin net): bf16 storage WITHOUT the folded heads (moda_nerf_train_bwd folds only n_out <= 32)
    ("skin36", dict(D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=36, raw_feat=True), 128, 0),
    # ONE code row for every sample (the rest-pose code of the forward warp) at 5120 rows: the row-bias gradients are plain
    # column sums over all samples (the wide-load column-sum kernel on bf16 rows)
    ("skin_rest", dict(D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=25, raw_feat=True), 128, 0)])
def test_nerf_fn_bf16_storage_route_matches_the_fp32_storage_route(name, kw, code_c, dir_c, monkeypatch):
    """The two backward routes of the bf16 throughput mode on one network: activations / backward tensors held as bf16 with
    the heads folded through T = dzd^T h (moda_nerf_train_bwd `folded`: no xyz_encoding_final output, bf16 weight copies,
    gemm_bf16.hip kernels) against fp32 storage with one GEMM per product (the route the rounding oracle pins).  Both round
    every GEMM operand to bf16; they differ in where (store vs load) and in the folded products being exact fp32, so they
    agree to the bf16 band, far inside the distance of either from exact fp32.  Also the ragged sizes: M = 259 rows."""
    from gpu_helpers import nerf_from_params
    R, S = (64, 80) if name == "skin_rest" else (7, 37)
    pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
    p = synth.nerf_params(52, "nfb/" + name, **pk)
    xyz = np.float32(0.3) * synth.normal(52, "nfb/xyz", (R, S, 3))
    code = synth.normal(52, "nfb/code", (1 if name == "skin_rest" else R, code_c)) if code_c else None
    dirs = synth.normal(52, "nfb/dir", (R, dir_c)) if dir_c else None
    n_out = kw["out_channels"] + (0 if kw["raw_feat"] else 1)
    gout = synth.normal(52, "nfb/g", (R, S, n_out))
    emb = moda_amd.Embedding(3, 10)

    def run(store):
        monkeypatch.setattr(A, "TRAIN_BF16_STORE", store)
        m = nerf_from_params(p, **kw).train()
        xg = T(xyz).requires_grad_(True)
        cg = None if code is None else T(code).requires_grad_(True)
        dg = None if dirs is None else T(dirs).requires_grad_(True)
        moda_amd.set_train_precision("bf16")
        try:
            yg = m.train_forward(xg, emb, code=cg, dir_src=dg)
            (yg * T(gout)).sum().backward()
        finally:
            moda_amd.set_train_precision("fp32")
        out = {"y": yg.detach(), "d_xyz": xg.grad}
        if cg is not None:
            out["d_code"] = cg.grad
        if dg is not None:
            out["d_dir"] = dg.grad
        out.update({pn: pt.grad for pn, pt in m.named_parameters() if pt.grad is not None})
        return out

    from helpers import rel_l2
    a, b = run(True), run(False)
    assert a.keys() == b.keys()
    assert torch.equal(a["y"], b["y"])                   # the forward is the same fused launch either way
    worst = ("", 0.0)
    for k in a:
        e = rel_l2(np_(a[k]), np_(b[k]))
        if e > worst[1]:
            worst = (k, e)
        assert e < 2e-2, (k, e)
    print(f"bf16 storage vs fp32 storage ({name}): worst rel-L2 over outputs and gradients {worst}")


@pytest.mark.parametrize("name,kw,code_c,dir_c", [
    ("coarse", dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False), 0, 91),
    ("skin", dict(D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=25, raw_feat=True), 128, 0),
    ("feat", dict(D=5, W=128, in_channels_xyz=63, in_channels_dir=0, out_channels=16, raw_feat=True), 0, 0)])
@pytest.mark.parametrize("R,S", [(7, 37), (33, 128)])
def test_fused_forward_bf16_dumps_are_the_rounded_fp32_dumps(name, kw, code_c, dir_c, R, S, monkeypatch):
    """moda_mlp_dump_fwd writes every hidden layer's activations for the backward: fp32 rows, or -- MODA_MLP_DUMP_BF16 -- bf16 rows
    through a lane-pair swap (8 x 256) or a per-wave LDS transpose (narrower nets).  Element for element the bf16 dump must
    be the fp32 dump rounded to nearest even (both come from the same accumulator), for whole and ragged tiles (259 rows)."""
    from gpu_helpers import nerf_from_params
    pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
    p = synth.nerf_params(53, "dmp/" + name, **pk)
    xyz = np.float32(0.3) * synth.normal(53, "dmp/xyz", (R, S, 3))
    code = synth.normal(53, "dmp/code", (R, code_c)) if code_c else None
    dirs = synth.normal(53, "dmp/dir", (R, dir_c)) if dir_c else None
    emb = moda_amd.Embedding(3, 10)
    M, W, D = R * S, kw["W"], kw["D"]

    def run(store):
        monkeypatch.setattr(A, "TRAIN_BF16_STORE", store)
        m = nerf_from_params(p, **kw).train()
        moda_amd.set_train_precision("bf16")
        try:
            y = m.train_forward(T(xyz).requires_grad_(True), emb, code=None if code is None else T(code),
                                dir_src=None if dirs is None else T(dirs))
        finally:
            moda_amd.set_train_precision("fp32")
        fn = y.grad_fn                    # walk back through the output views to the NerfFn node
        while fn is not None and "NerfFn" not in type(fn).__name__:
            fn = fn.next_functions[0][0]
        ws = fn.saved_tensors[3]
        return y.detach(), ws

    y32, ws32 = run(False)
    y16, ws16 = run(True)
    assert torch.equal(y32, y16)
    off = M * 64                                   # the positional encoding (M x 64 fp32) comes first in the workspace
    for l in range(D):
        a = ws32[off + l * M * W: off + (l + 1) * M * W].view(M, W)
        b = ws16[off + l * M * W: off + (l + 1) * M * W].view(torch.bfloat16)[:M * W].view(M, W)
        assert torch.equal(a.bfloat16(), b), (name, l, (a.bfloat16().float() - b.float()).abs().max().item())
        assert float(a.abs().max()) > 0


@pytest.mark.parametrize("tag", ["ana", "fd"])
def test_eikonal_loss_matches_reference(tag):
    """eikonal_loss (loss_utils.py:73-104): loss and parameter gradients against the reference's (g14); the analytic
    form's reverse sweep, written as forward nodes, must reproduce the reference's double backward."""
    from test_torch_ref import G14, G14_GRADS, G14_TOL
    g = golden("g14_eikonal")
    models, emb = make_models(14, 0)
    coarse = models["coarse"].train()
    pts = T(np.float32(G14["scale"]) * synth.normal(14, "g14/pts", G14["shape"]))
    gr, sig = moda_amd.nerf_gradient(coarse, emb["xyz"], pts.view(1, -1, 3), sigma_only=True)
    assert rel_err(np_(gr), g["grad"]) < 1e-4 and rel_err(np_(sig), g["sigmas"]) < 1e-4
    loss = moda_amd.eikonal_loss(coarse, emb["xyz"], pts, G14["bound"], tag == "fd")
    loss.backward()
    tol = G14_TOL[tag]
    assert abs(float(loss.detach()) - float(g[tag + "_loss"])) < tol * abs(float(g[tag + "_loss"]))
    sd = dict(coarse.named_parameters())
    for k in G14_GRADS:
        ref = g[f"{tag}_d_{k}"]
        got = np_(sd[k].grad) if sd[k].grad is not None else np.zeros_like(ref)
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, k
        else:
            l2 = float(np.linalg.norm(got.astype(np.float64) - ref) / np.linalg.norm(ref.astype(np.float64)))
            assert l2 < 5 * tol, (k, l2)


def test_eikonal_loss_subsamples_rays():
    """More than 1000 rays: the ray subset is injected (`eik_inds`) and the loss equals the loss of that subset."""
    models, emb = make_models(14, 0)
    coarse = models["coarse"].train()
    pts = T(np.float32(0.1) * synth.normal(14, "eik/pts", (1100, 4, 3)))
    inds = torch.from_numpy(np.random.default_rng(3).permutation(1100)[:1000].copy())
    a = moda_amd.eikonal_loss(coarse, emb["xyz"], pts, [0.2, 0.2, 0.2], False, rng={"eik_inds": inds})
    b = moda_amd.eikonal_loss(coarse, emb["xyz"], pts[inds.to(DEV)], [0.2, 0.2, 0.2], False)
    assert abs(float(a.detach()) - float(b.detach())) < 1e-6 * abs(float(b.detach()))


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_residual_displacement_field_matches_reference(mode):
    """nerf_dis on (moda.py:80 switched on): backward warp minus the field at the frame-space samples, forward warps of
    the displaced canonical samples (cycle, target, dense target; skinning weights still taken at the undisplaced
    points), dis_reg / dis_reg_forward -- outputs and gradients against the reference's (g15)."""
    from test_torch_ref import G15_OUT, G15_LOSS, G15_LEAVES, G15_PARAMS
    g = golden("g15_dis_" + mode)
    N, S, B = 48, 12, 25
    models, emb = make_models(15, B, with_skin=True, perturb_bones=True, with_dis=True)
    train = mode == "train"
    for m in models.values():
        if isinstance(m, torch.nn.Module):
            m.train(train)
    if train:
        models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
    rays = rays_to_gpu(synth.make_rays(15, N, B, rays_per_frame=8))
    rays.update(rays_to_gpu(synth.make_corresp_rays(15, N, B, rays_per_frame=8)))
    if train:
        for k in G15_LEAVES:
            rays[k].requires_grad_(True)
    with (torch.enable_grad() if train else torch.no_grad()):
        res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, img_size=512,
                                   opts=make_opts(dist_corresp=True))
    for k in G15_OUT + ("flo_coarse", "fdp_coarse"):
        assert rel_err(np_(res[k]), g[k]) < 1e-4, (k, rel_err(np_(res[k]), g[k]))
    for k in ("flo_valid", "fdp_valid"):
        assert np.array_equal(np_(res[k]), g[k])
    if not train:
        return
    loss = 0
    for k in G15_LOSS:
        loss = loss + (T(synth.normal(15, "g15/c/" + k, tuple(res[k].shape))) * res[k]).sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    for k in G15_LEAVES:
        check_grad("d_" + k, np_(rays[k].grad), g, 2e-3, l2=True)
    check_grad("d_bones_rst", np_(models["bones_rst"].grad), g, 2e-3, l2=True)
    check_grad("d_rest_pose_code", np_(models["rest_pose_code"].weight.grad), g, 2e-3, l2=True)
    for mn, pn in G15_PARAMS:
        check_grad(f"d_{mn}.{pn}", np_(dict(models[mn].named_parameters())[pn].grad), g, 2e-3, l2=True)


def test_rgb_filter_matches_reference():
    """opts.rgb_filter (rendering.py:171, 225-230): outputs and gradients against the reference's (g16); the semantic
    weight adds a direct path from the rendered colour into sigma."""
    g = golden("g16_rgb_filter")
    N, S, B = 48, 12, 25
    models, emb = make_models(16, B, with_skin=True, perturb_bones=True)
    for m in models.values():
        if isinstance(m, torch.nn.Module):
            m.train()
    rays = rays_to_gpu(synth.make_rays(16, N, B, rays_per_frame=8))
    leaves = ("rays_o", "rays_d", "bone_rts", "env_code")
    for k in leaves:
        rays[k].requires_grad_(True)
    res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, img_size=512, opts=make_opts(rgb_filter=True))
    loss = 0
    for k in ("img_coarse", "depth_rnd", "sil_coarse"):
        assert rel_err(np_(res[k]), g[k]) < 1e-4, (k, rel_err(np_(res[k]), g[k]))
        loss = loss + (T(synth.normal(16, "g16/c/" + k, tuple(res[k].shape))) * res[k]).sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    for k in leaves:
        check_grad("d_" + k, np_(rays[k].grad), g, 2e-3, l2=True)
    sd = dict(models["coarse"].named_parameters())
    for k in ("sigma.weight", "sigma.bias", "rgb.0.weight", "xyz_encoding_8.0.weight", "beta"):
        check_grad("d_coarse." + k, np_(sd[k].grad), g, 2e-3, l2=True)
    # and the no-grad (fused) route gives the same picture
    with torch.no_grad():
        ev = moda_amd.render_rays(models, emb, {k: v.detach() for k, v in rays.items()}, N_samples=S, noise_std=0.0,
                                  img_size=512, opts=make_opts(rgb_filter=True))
    assert rel_err(np_(ev["img_coarse"]), g["img_coarse"]) < 1e-4


@pytest.mark.parametrize("mode,use_ot", [("ot", True), ("softmax", False)])
def test_back_correspondence_term_matches_reference(mode, use_ot):
    """opts.use_corr (off by default, moda.py:157): corr_err = |P P^T - I|_2 per pixel on the matching probabilities
    (loss_utils.py:386-391), whose gradient re-enters the hand-derived Sinkhorn / softmax backward as an upstream
    gradient on P -- outputs and nerf_feat gradients against the reference's (g17)."""
    from test_torch_ref import rel_l2
    g = golden("g17_corr_" + mode)
    N, S, B = 48, 12, 25
    models, emb = make_models(17, B, with_skin=True, with_feat=True, perturb_bones=True)
    for m in models.values():
        if isinstance(m, torch.nn.Module):
            m.train()
    rays = rays_to_gpu(synth.make_rays(17, N, B, rays_per_frame=8))
    rays.update(rays_to_gpu(synth.make_corresp_rays(17, N, B, rays_per_frame=8)))
    rays.update(rays_to_gpu(synth.make_feat_rays(17, N, rays_per_frame=8)))
    res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, img_size=512, obj_bound=G11_BOUND,
                               opts=make_opts(dist_corresp=True, use_corresp=True, use_ot=use_ot, use_corr=True),
                               rng={"feat_noise": T(g["rng_randn_like"])})
    keys = ("pts_pred", "feat_err", "corr_err", "proj_err")
    loss = 0
    for k in keys:
        got = np_(res[k].float())
        assert got.shape == g[k].shape, (k, got.shape, g[k].shape)
        assert rel_err(got, g[k]) < 2e-4, (mode, k, rel_err(got, g[k]))
        loss = loss + (T(synth.normal(17, "g17/c/" + k, tuple(res[k].shape))) * res[k]).sum()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-4 * abs(float(g["loss"]))
    loss.backward()
    sd = dict(models["nerf_feat"].named_parameters())
    for pn in ("rgb.0.weight", "xyz_encoding_1.0.weight") + (() if use_ot else ("beta",)):
        assert sd[pn].grad is not None, pn
        assert rel_l2(np_(sd[pn].grad), g["d_nerf_feat." + pn]) < 1e-2, (pn, rel_l2(np_(sd[pn].grad), g["d_nerf_feat." + pn]))


def test_uncertainty_head_and_appearance_code_train_route():
    """G19 train mode: `unc_pred` and the appearance-code columns under autograd -- outputs and gradients (ray leaves,
    nerf_unc and colour-branch parameters) against the reference's own autograd."""
    from gpu_helpers import unc_models
    g = golden("g19_unc_app_train")
    N, S, B = 48, 12, 25
    models, emb = unc_models(19)
    for m in models.values():
        if isinstance(m, torch.nn.Module):
            m.train()
    rays = rays_to_gpu(synth.make_rays(19, N, B, rays_per_frame=8, with_app=True))
    rays.update(rays_to_gpu(synth.make_unc_rays(19, N, 8)))
    leaves = ("rays_o", "rays_d", "bone_rts", "env_code", "appearance_code", "vid_code", "ts", "xysn")
    for k in leaves:
        rays[k].requires_grad_(True)
    res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
    loss = 0
    for k in ("img_coarse", "sil_coarse", "depth_rnd", "unc_pred", "frame_cyc_dis"):
        assert rel_err(np_(res[k]), g[k]) < 1e-4, (k, rel_err(np_(res[k]), g[k]))
        loss = loss + (T(synth.normal(19, "g19/c/" + k, tuple(res[k].shape))) * res[k]).sum()
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    # heads that do not pass through the density gain are tight; the rest carries the 576-sample ReLU-switch band of G9
    for k in ("vid_code", "ts", "xysn"):
        check_grad("d_" + k, np_(rays[k].grad), g, 5e-4, l2=True)
    for k in ("rays_o", "rays_d", "bone_rts", "env_code", "appearance_code"):
        check_grad("d_" + k, np_(rays[k].grad), g, 1e-2, l2=True)
    for mn, pn, tol in (("nerf_unc", "rgb.0.weight", 5e-4), ("nerf_unc", "xyz_encoding_1.0.weight", 5e-4),
                        ("nerf_unc", "dir_encoding.0.weight", 5e-4), ("nerf_unc", "xyz_encoding_8.0.bias", 5e-4),
                        ("coarse", "dir_encoding.0.weight", 1e-2), ("coarse", "rgb.0.weight", 1e-2), ("coarse", "sigma.weight", 1e-2)):
        check_grad(f"d_{mn}.{pn}", np_(dict(models[mn].named_parameters())[pn].grad), g, tol, l2=True)


def test_render_rays_gradients_large_fixture_1e3(parity_precision):
    """G21 (512 rays x 64 samples = 32768 samples): end-to-end gradients through moda_amd.render_rays vs the REFERENCE's
    autograd at <= 1e-3 relative L2 per tensor -- the bar G9's 576-sample fixture cannot support (one ReLU switch there is
    2e-3 of a gradient norm)."""
    from test_torch_ref import check_large_grads
    g = golden("g21_grad_large")
    N, S, B = 512, 64, 25
    models, emb = make_models(21, B, with_skin=True, perturb_bones=True)
    for m in models.values():
        if isinstance(m, torch.nn.Module):
            m.train()
    models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
    models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
    rays = rays_to_gpu(synth.make_rays(21, N, B, rays_per_frame=32))
    for k in GRAD_LEAVES:
        rays[k].requires_grad_(True)
    res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
    loss = 0
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "frame_cyc_dis"):
        loss = loss + (T(synth.normal(21, "g21/c/" + k, tuple(res[k].shape))) * res[k]).sum()
    for k in ("img_coarse", "sil_coarse", "frame_cyc_dis"):
        assert rel_err(np_(res[k]), g[k]) < 1e-4, (k, rel_err(np_(res[k]), g[k]))
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    grads = {"d_" + k: rays[k].grad for k in GRAD_LEAVES}
    grads.update({f"d_coarse.{pn}": p.grad for pn, p in models["coarse"].named_parameters()})
    grads.update({f"d_nerf_skin.{pn}": p.grad for pn, p in models["nerf_skin"].named_parameters()})
    grads.update({"d_bones_rst": models["bones_rst"].grad, "d_skin_aux": models["skin_aux"].grad,
                  "d_rest_pose_code": models["rest_pose_code"].weight.grad})
    worst = check_large_grads(g, grads, 1e-3)
    print("G21 worst relative L2 gradient error vs the reference:", worst)
    assert worst[1] < 1e-3, worst
    assert_gradients_within_f64_truth("g21_grad_large", {k: np_(v) for k, v in grads.items() if v is not None}, parity_precision, N, S)


def test_render_rays_gradients_flip_free_rays_against_float64_truth(parity_precision):
    """Every gradient tensor -- parameters included -- against the float64 truth with ReLU coin flips taken out of the
    question: 256 rays x 64 samples, the loss weights of every ray that holds a ReLU pre-activation within 2e-6 (relative) of
    zero in float64 set to zero, so that no sample whose gradient is undetermined at fp32 accuracy contributes anywhere (rays
    are independent).  Truth: oracle/torch_ref.py evaluated in float64 on this box -- pinned to the reference run in float64
    at 1e-10 by tests/test_torch_ref.py.  Bar: 5e-5 relative L2 on EVERY tensor, in both parity-grade precisions (observed: worst
    1.8e-5 on the skin network's bias gradients -- column sums of signed terms, fp32 atomics in run-to-run order -- median 1.3e-6)."""
    from test_torch_ref import torch_scene
    from helpers import rel_l2, TRUTH_FLOOR, AMB_TOL
    N, S, B, seed = 256, 64, 25, 27
    # ---- float64 truth + conditioning
    m = torch_scene(seed, B, True, perturb_bones=True, requires_grad=True, dtype=torch.float64)
    rays_c = {k: TC(v).double() for k, v in synth.make_rays(seed, N, B, rays_per_frame=32).items()}
    for k in GRAD_LEAVES:
        rays_c[k].requires_grad_(True)
    tr.RELU_MARGINS = []
    try:
        res_c = tr.render_rays(m, rays_c, S)
        margins = tr.RELU_MARGINS
    finally:
        tr.RELU_MARGINS = None
    assert margins and all(mg.numel() == N * S for mg in margins)
    clean = torch.stack([mg.reshape(N, S).min(1).values for mg in margins]).min(0).values >= AMB_TOL       # (N,) rays without an undetermined ReLU
    n_clean = int(clean.sum())
    assert 32 <= n_clean < N, n_clean
    cs = {k: synth.normal(seed, "ff/c/" + k, tuple(res_c[k].shape)) * clean.numpy().astype(np.float32).reshape((N,) + (1,) * (res_c[k].dim() - 1))
          for k in ("img_coarse", "depth_rnd", "sil_coarse", "frame_cyc_dis")}
    sum((TC(c).double() * res_c[k]).sum() for k, c in cs.items()).backward()
    truth = {"d_" + k: rays_c[k].grad.numpy() for k in GRAD_LEAVES}
    truth.update({f"d_{mn}.{pn}": p.grad.numpy() for mn in ("coarse", "nerf_skin") for pn, p in m[mn].items() if p.grad is not None})
    truth.update({"d_bones_rst": m["bones_rst"].grad.numpy(), "d_skin_aux": m["skin_aux"].grad.numpy(), "d_rest_pose_code": m["rest_pose_code"].grad.numpy()})
    # ---- HIP path
    models, emb = make_models(seed, B, with_skin=True, perturb_bones=True)
    for mm in models.values():
        if isinstance(mm, torch.nn.Module):
            mm.train()
    models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
    models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
    rays = rays_to_gpu(synth.make_rays(seed, N, B, rays_per_frame=32))
    for k in GRAD_LEAVES:
        rays[k].requires_grad_(True)
    res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
    sum((T(c) * res[k]).sum() for k, c in cs.items()).backward()
    got = {"d_" + k: np_(rays[k].grad) for k in GRAD_LEAVES}
    got.update({f"d_{mn}.{pn}": np_(p.grad) for mn in ("coarse", "nerf_skin") for pn, p in models[mn].named_parameters() if p.grad is not None})
    got.update({"d_bones_rst": np_(models["bones_rst"].grad), "d_skin_aux": np_(models["skin_aux"].grad),
                "d_rest_pose_code": np_(models["rest_pose_code"].weight.grad)})
    assert set(got) == set(truth), set(got) ^ set(truth)
    errs = sorted(((rel_l2(got[k], truth[k]), k) for k in truth), reverse=True)
    print(f"flip-free gradients [{parity_precision}], {n_clean} of {N} rays carry weight: worst " + ", ".join(f"{k}={e:.1e}" for e, k in errs[:4])
          + f"; median {np.median([e for e, _ in errs]):.1e}")
    assert errs[0][0] < 5e-5 and np.median([e for e, _ in errs]) < TRUTH_FLOOR / 4, errs[:4]


@pytest.mark.parametrize("store", ["bf16_store", "fp32_store"])
def test_bf16_training_mode_against_its_rounding_oracle(store, monkeypatch):
    """Throughput mode of the training route (moda_amd.set_train_precision('bf16'): every GEMM's operands rounded to bf16,
    fp32 products / sums / master weights / activations / gradients) against the CPU restatement with the SAME operand
    rounding in forward and backward (oracle/torch_ref.py LINEAR_BF16_OPERANDS), on the G21 inputs (512 rays x 64
    samples); and its distance from the exact-fp32 reference gradients, for the record.  Both storage forms of the saved
    activations / backward tensors (MODA_TRAIN_BF16_STORE: bf16, the default with the fused forward; fp32) meet the same
    bounds: a GEMM operand is rounded to bf16 on its way into the MFMA anyway, the bf16 store only moves that rounding."""
    monkeypatch.setattr(A, "TRAIN_BF16_STORE", store == "bf16_store")
    from test_torch_ref import torch_scene
    g = golden("g21_grad_large")
    N, S, B = 512, 64, 25

    def loss_of(res, conv):
        tot = 0
        for k in ("img_coarse", "depth_rnd", "sil_coarse", "frame_cyc_dis"):
            tot = tot + (conv(synth.normal(21, "g21/c/" + k, tuple(res[k].shape))) * res[k]).sum()
        return tot

    # rounding oracle (CPU)
    tr.LINEAR_BF16_OPERANDS = True
    try:
        m = torch_scene(21, B, True, perturb_bones=True, requires_grad=True)
        rays_c = {k: TC(v) for k, v in synth.make_rays(21, N, B, rays_per_frame=32).items()}
        for k in GRAD_LEAVES:
            rays_c[k].requires_grad_(True)
        res_c = tr.render_rays(m, rays_c, S)
        loss_of(res_c, TC).backward()
    finally:
        tr.LINEAR_BF16_OPERANDS = False
    # HIP path
    models, emb = make_models(21, B, with_skin=True, perturb_bones=True)
    for mm in models.values():
        if isinstance(mm, torch.nn.Module):
            mm.train()
    models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
    models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
    rays = rays_to_gpu(synth.make_rays(21, N, B, rays_per_frame=32))
    for k in GRAD_LEAVES:
        rays[k].requires_grad_(True)
    moda_amd.set_train_precision("bf16")
    calls = []
    orig_call = A.L.call

    def spy(name, *a):
        calls.append(name)
        return orig_call(name, *a)
    A.L.call = spy
    try:
        res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
        loss_of(res, T).backward()
    finally:
        A.L.call = orig_call
        moda_amd.set_train_precision("fp32")
    # the throughput mode runs each network's forward as ONE fused launch (coarse + two skin evaluations here)
    assert calls.count("moda_nerf_train_fwd_fused") == 3 and "moda_nerf_train_fwd" not in calls, calls.count("moda_nerf_train_fwd_fused")
    from helpers import rel_l2
    for k in ("img_coarse", "sil_coarse", "frame_cyc_dis"):
        e = rel_err(np_(res[k]), res_c[k].detach().numpy())
        assert e < 2e-3, (k, e)                                    # same rounding, different summation order / ReLU ties
        assert rel_err(np_(res[k]), g[k]) < 3e-2, k                # vs the exact-fp32 reference: the bf16 band
    worst, worst_ref = ("", 0.0), ("", 0.0)
    pairs = [("d_" + k, rays[k].grad, rays_c[k].grad) for k in GRAD_LEAVES]
    pairs += [(f"d_coarse.{pn}", p.grad, m["coarse"][pn].grad) for pn, p in models["coarse"].named_parameters()]
    pairs += [(f"d_nerf_skin.{pn}", p.grad, m["nerf_skin"][pn].grad) for pn, p in models["nerf_skin"].named_parameters()]
    pairs += [("d_bones_rst", models["bones_rst"].grad, m["bones_rst"].grad), ("d_skin_aux", models["skin_aux"].grad, m["skin_aux"].grad)]
    for name, a, b in pairs:
        if a is None or b is None:
            assert a is None and b is None, name
            continue
        e = rel_l2(np_(a), b.numpy())
        if e > worst[1]:
            worst = (name, e)
        # the two sides round the same operands, but an activation that lands on a bf16 rounding boundary can round either
        # way after a different summation order (a 4e-3 relative step): noise that shows most in gradients with heavy
        # cancellation -- skin_aux[0], ONE scalar that sums every sample's contribution with both signs, moves by ~10 %
        # between two bf16 evaluations that differ only in where they round (per-layer GEMMs vs the fused forward kernel with
        # its folded final layer and hardware-sine encoding; vs the exact-fp32 reference it is 9 % off either way)
        # (measured worst of the rest: 0.049 on d_nerf_skin.xyz_encoding_3.0.bias, a 64-entry bias gradient -- sums over every
        #  sample with both signs -- in both storage forms and across boxes; 6e-2 leaves it a margin against the run-to-run
        #  order of the split-K atomics)
        assert e < (2e-1 if name == "d_skin_aux" else 6e-2), (name, e)
        if name in g:
            er = rel_l2(np_(a), g[name])
            if er > worst_ref[1]:
                worst_ref = (name, er)
    print(f"bf16 training mode ({store}): worst rel-L2 gradient error vs its rounding oracle {worst}, vs the fp32 reference {worst_ref}")
    assert worst_ref[1] < 0.2, worst_ref


def test_gemm_row_count_beyond_the_grid_y_limit():
    """moda_gemm_f32_ex launches its tiles on a linear grid: M = 9,000,000 rows (70,313 row tiles, more than a 65,535-wide
    grid.y holds) runs instead of failing at launch; moda_colsum_f32 folds its row blocks the same way."""
    M, K, N = 9_000_000, 8, 8
    a = torch.rand(M, K, device=DEV)
    b = torch.rand(K, N, device=DEV)
    got = A.gemm(a, b)
    idx = torch.tensor([0, 1, 127, 128, 8_388_607, 8_388_608, M - 1], device=DEV)
    want = a[idx].double() @ b.double()
    assert rel_err(np_(got[idx]), want.cpu().numpy()) < 1e-6
    s = torch.zeros(N, device=DEV)
    from moda_amd import _lib as L
    L.call("moda_colsum_f32", L.ptr(got), M, N, N, L.ptr(s), L.stream())
    assert rel_err(np_(s), got.double().sum(0).cpu().numpy()) < 1e-4


@pytest.mark.parametrize("N", [64, 256, 37])
@pytest.mark.parametrize("M", [1000, 40_000])
def test_gemm_bf16_storage_types(M, N):
    """MODA_GEMM_{A,B,C,MASK}_BF16 (moda_hip.h; bf16 mode only): operands held as bf16 in memory give the result of the same
    bf16-mode GEMM on their fp32 widenings; a bf16 C is that result rounded to nearest even.  Forms of the training backward:
    dX = dZ @ W with the ReLU mask (k-fast A, k-slow fp32 B), dW += dZ^T @ X with the bias sums (m-fast A, split-K atomics
    into fp32), C += (form 2), the fp32 d_pe store.  N = 64 / 256 take the bf16-native kernels of gemm_bf16.hip, N = 37 the
    generic kernel with typed element access (moda_g3_try declines it); both must agree with the same reference."""
    from moda_amd import _lib as L
    torch.manual_seed(5)
    K = 96
    a = torch.randn(M, K, device=DEV).bfloat16()
    w = torch.randn(K, N, device=DEV)                     # fp32 weights, k-slow (sbn == 1): rounded to bf16 by the kernel
    mask = torch.randn(M, N, device=DEV).bfloat16()
    c0 = torch.randn(M, N, device=DEV).bfloat16()

    def run(A_, sam, sak, B_, sbk, sbn, C_, Mm, Nn, Kk, flags, mask_=None, acc=0, split=1, asum=None):
        d = L.GemmDesc(A=A_.data_ptr(), sam=sam, sak=sak, A2=None, sam2=0, K1=Kk, B=B_.data_ptr(), sbk=sbk, sbn=sbn,
                       C=C_.data_ptr(), ldc=C_.stride(0), M=Mm, N=Nn, K=Kk, bias=None, rowbias=None, ld_rowbias=0,
                       rows_per_bias=1, mask_src=None if mask_ is None else mask_.data_ptr(),
                       ld_mask=0 if mask_ is None else mask_.stride(0), act=0, accumulate=acc, split_k=split, reserved=flags,
                       a_sum=None if asum is None else asum.data_ptr())
        L.call("moda_gemm_f32_ex", L._c.byref(d), L.stream())

    BF, FA, FB, FC, FM = 1, 2, 4, 8, 16
    want = a.float().double() @ w.bfloat16().float().double()
    scale = want.abs().max().item()

    def close_bf16(got, ref):          # one bf16 rounding step of the largest magnitude around: sums differ in order only
        assert (got.float() - ref.float()).abs().max().item() <= 2 ** -7 * scale

    # dX form: bf16 dZ, fp32 W, masked bf16 result
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    run(a, K, 1, w, N, 1, out, M, N, K, BF | FA | FC | FM, mask_=mask)
    ref = torch.where(mask.float() > 0, want, torch.zeros_like(want)).float().bfloat16()
    close_bf16(out, ref)
    assert torch.equal(out.float() == 0, ref.float() == 0) or (mask.float() > 0).all()     # the mask zeroes exactly its entries
    assert rel_err(np_(out.float()), np_(ref.float())) < 4e-3
    # C = C + A @ B on a bf16 C, no mask
    out = c0.clone()
    run(a, K, 1, w, N, 1, out, M, N, K, BF | FA | FC, acc=2)
    close_bf16(out, (c0.float().double() + want).float().bfloat16())
    # fp32 result (the d_pe product), plain and +=
    out32 = torch.empty(M, N, device=DEV)
    run(a, K, 1, w, N, 1, out32, M, N, K, BF | FA)
    assert rel_err(np_(out32), want.cpu().numpy()) < 1e-5
    run(a, K, 1, w, N, 1, out32, M, N, K, BF | FA, acc=2)
    assert rel_err(np_(out32), 2 * want.cpu().numpy()) < 1e-5
    # dW (K x N) += dZ^T @ X over the M rows: m-fast bf16 dZ, k-slow X (bf16, and fp32 as the positional encoding is),
    # split-K atomics and the bias sums in fp32
    x = torch.randn(M, N, device=DEV).bfloat16()
    wantW = (a.float().double().T @ x.float().double()).cpu().numpy()
    for xx, fl in ((x, FB), (x.float(), 0)):
        dW = torch.zeros(K, N, device=DEV)
        db = torch.zeros(K, device=DEV)
        run(a, 1, K, xx, N, 1, dW, K, N, M, BF | FA | fl, acc=1, split=3, asum=db)
        assert rel_err(np_(dW), wantW) < 1e-5, fl
        assert rel_err(np_(db), a.float().double().sum(0).cpu().numpy()) < 1e-5, fl
    # the positional-encoding shape: 63 columns in rows of 64
    if N == 64:
        pe = torch.randn(M, 64, device=DEV)
        dW = torch.zeros(K, 63, device=DEV)
        run(a, 1, K, pe, 64, 1, dW, K, 63, M, BF | FA, acc=1, split=2)
        assert rel_err(np_(dW), (a.float().double().T @ pe[:, :63].bfloat16().float().double()).cpu().numpy()) < 1e-5
    # refused: bf16 C with the atomics form; storage types outside the bf16 mode
    with pytest.raises(Exception):
        run(a, 1, K, x, N, 1, torch.zeros(K, N, device=DEV).bfloat16(), K, N, M, BF | FA | FB | FC, acc=1, split=2)
    with pytest.raises(Exception):
        run(a, K, 1, w, N, 1, out32, M, N, K, FA)


@pytest.mark.parametrize("case,N,S", [("small", 48, 8), ("large", 1100, 4)])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_s3im_loss_matches_reference(case, N, S, mode):
    """opts.s3im_loss (default off, moda.py:170; rendering.py:528-532, 566-567; loss_utils.py:575-702 S3IM / SSIM): the term
    itself within 1e-4 of the reference with the reference's permutations injected, the masked img_coarse / img_at_samp the
    reference returns with the flag on (S3IM.forward masks its arguments in place), and in train mode the gradients of
    3 * s3im_loss + <c, img_loss_samp> within 1e-3 relative L2 (tests/golden/g23_s3im_*.npz).  N < 1024: rows repeated;
    N > 1024: the first 1024 rows."""
    from test_torch_ref import rel_l2
    g = golden(f"g23_s3im_{case}_{mode}")
    train = mode == "train"
    B = 25
    models, emb = make_models(23, B, with_skin=True, perturb_bones=True)
    if train:
        models["coarse"].train()
    rays = rays_to_gpu(synth.make_rays(23, N, B, rays_per_frame=4))
    rays.update(rays_to_gpu(synth.make_corresp_rays(23, N, B, rays_per_frame=4)))
    observed = rays["img_at_samp"].clone()
    leaves = (("rays_d", "bone_rts", "time_embedded", "env_code") if case == "small" else ("rays_d",)) if train else ()
    for k in leaves:
        rays[k].requires_grad_(True)
    with (torch.enable_grad() if train else torch.no_grad()):
        res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, img_size=512,
                                   opts=make_opts(dist_corresp=True, use_corresp=True, s3im_loss=True),
                                   rng={"s3im_perms": T(g["perms"])})
    e = abs(float(res["s3im_loss"]) - float(g["s3im_loss"])) / abs(float(g["s3im_loss"]))
    print(f"s3im {case}/{mode}: {float(res['s3im_loss']):.7f} vs reference {float(g['s3im_loss']):.7f} ({e:.1e})")
    assert e < 1e-4
    for k in ("img_coarse", "img_at_samp", "img_loss_samp", "sil_coarse"):
        assert rel_err(np_(res[k].float()), g[k]) < 1e-4, (k, rel_err(np_(res[k].float()), g[k]))
    # the caller's observed colours are masked in place, as the reference leaves them (loss_utils.py:666)
    assert torch.equal(rays["img_at_samp"], observed * rays["sil_at_samp"])
    if train:
        c = T(synth.normal(23, "g23/c/img", tuple(res["img_loss_samp"].shape)))
        loss = 3.0 * res["s3im_loss"] + (c * res["img_loss_samp"]).sum()
        assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
        loss.backward()
        got = {"d_" + k: rays[k].grad for k in leaves}
        for pn in ("rgb.0.weight", "sigma.weight", "xyz_encoding_1.0.weight", "dir_encoding.0.weight"):
            got["d_coarse." + pn] = dict(models["coarse"].named_parameters())[pn].grad
        got["d_coarse.beta"] = models["coarse"].beta.grad
        for k, v in got.items():
            assert v is not None, k
            err = rel_l2(np_(v), g[k])
            print(f"   {k}: rel-L2 {err:.2e}")
            assert err < 1e-3, (k, err)


def test_s3im_kernel_against_the_torch_restatement():
    """moda_s3im alone (forward value and d/d rgb) against oracle/torch_ref.s3im_loss in float64, on random colours, a soft
    mask and random permutations, at N below and above the 1024-row patch."""
    from oracle import torch_ref as tr
    from moda_amd import loss_utils as LU
    for N in (300, 1024, 2500):
        src = synth.uniform(31, f"s3/src{N}", (N, 3))
        tar = np.clip(src + np.float32(0.1) * synth.normal(31, f"s3/tar{N}", (N, 3)), 0, 1).astype(np.float32)
        mask = (synth.uniform(31, f"s3/m{N}", (N, 1)) < 0.7).astype(np.float32) * synth.uniform(31, f"s3/mm{N}", (N, 1))
        gen = torch.Generator().manual_seed(31 + N)
        perms = torch.stack([torch.randperm(1024, generator=gen) for _ in range(9)])
        a = T(src).requires_grad_(True)
        out = LU.s3im_loss(a, T(tar), T(mask), rng={"s3im_perms": perms})
        out.backward()
        ac = TC(src).double().requires_grad_(True)
        ref = tr.s3im_loss(ac, TC(tar).double(), TC(mask).double(), perms)
        ref.backward()
        assert abs(float(out) - float(ref)) < 2e-5 * abs(float(ref)), (N, float(out), float(ref))
        err = float((a.grad.cpu().double() - ac.grad).norm() / ac.grad.norm())
        print(f"s3im kernel N={N}: value {float(out):.6f} vs {float(ref):.6f}, gradient rel-L2 {err:.2e}")
        assert err < 1e-4, (N, err)


def test_ray_loss_and_masked_mean_kernels_against_torch():
    """moda_ray_loss (the img / sil / flo terms of rendering.py:518-571 with the silhouette class balance :535-539 and the
    confidence normalisation :549-556, one kernel each way) and moda_masked_mean (x[m].mean() of moda.py:540-640) against the
    same arithmetic written with torch ops as the reference writes it (boolean gathers and `if ... .sum() > 0`), values and
    gradients, training and eval, including the degenerate batches (no silhouette pixel; no valid flow pixel)."""
    from moda_amd import autograd as A, loss_utils as LU
    N = 777
    def ref(rgb, sil, flo, valid, img_at, sil_at, vis_at, flo_at, cfd_at, training):
        img = (rgb - img_at).pow(2).mean(-1)[..., None]
        if training and sil_at.sum() > 0 and (1 - sil_at).sum() > 0:
            pos_wt = vis_at.sum() / sil_at[vis_at > 0].sum()
            neg_wt = vis_at.sum() / (1 - sil_at[vis_at > 0]).sum()
            bal = 0.5 * pos_wt * sil_at + 0.5 * neg_wt * (1 - sil_at)
        else:
            bal = 1
        sl = (sil[..., None] - sil_at).pow(2) * bal * vis_at
        fl = (flo - flo_at).pow(2).sum(-1)
        sf = (sil_at > 0) & (valid == 1)
        sf[cfd_at == 0] = False
        cf = cfd_at
        if sf.sum() > 0:
            cf = cfd_at / cfd_at[sf].mean()
        fl = fl[..., None] * cf
        return img * sil_at, sl, fl * sil_at, sf
    for case in ("normal", "no_sil", "no_flow"):
        for training in (True, False):
            u = lambda tag, shape: synth.uniform(41, f"rl/{case}/{tag}", shape)
            inp = dict(rgb=u("rgb", (N, 3)), sil=u("sil", (N,)), flo=synth.normal(41, f"rl/{case}/flo", (N, 2)),
                       valid=(u("valid", (N, 1)) < 0.8).astype(np.float32), img_at=u("img", (N, 3)),
                       sil_at=(u("sa", (N, 1)) < 0.6).astype(np.float32), vis_at=(u("va", (N, 1)) < 0.9).astype(np.float32),
                       flo_at=synth.normal(41, f"rl/{case}/fa", (N, 2)), cfd_at=np.where(u("c", (N, 1)) < 0.2, 0, u("c2", (N, 1))).astype(np.float32))
            if case == "no_sil":
                inp["sil_at"][:] = 0
            if case == "no_flow":
                inp["valid"][:] = 0
            g = {k: T(v) for k, v in inp.items()}
            c = {k: TC(v).double() for k, v in inp.items()}
            for d in (g, c):
                for k in ("rgb", "sil", "flo"):
                    d[k].requires_grad_(True)
            out = A.RayLossFn.apply(g["rgb"], g["sil"], g["flo"], g["valid"], g["img_at"], g["sil_at"], g["vis_at"], g["flo_at"],
                                    g["cfd_at"], training)
            want = ref(*[c[k] for k in ("rgb", "sil", "flo", "valid", "img_at", "sil_at", "vis_at", "flo_at", "cfd_at")], training)
            assert torch.equal(out[3].cpu(), want[3])
            w = [TC(synth.normal(41, f"rl/{case}/w{i}", (N, 1))) for i in range(3)]
            lg = sum((o * wi.to(o.device)).sum() for o, wi in zip(out[:3], w))
            lc = sum((o * wi.double()).sum() for o, wi in zip(want[:3], w))
            lg.backward(); lc.backward()
            for i in range(3):
                assert rel_err(np_(out[i]), want[i].detach().numpy()) < 2e-6, (case, training, i)
            for k in ("rgb", "sil", "flo"):
                assert rel_err(np_(g[k].grad), c[k].grad.numpy()) < 2e-6, (case, training, k)
    # masked mean
    for k in (1, 3):
        x = synth.normal(42, f"mm/x{k}", (N, k))
        m = synth.uniform(42, f"mm/m{k}", (N, 1)) < 0.4
        xg, xc = T(x).requires_grad_(True), TC(x).double().requires_grad_(True)
        got = LU.masked_mean(xg, T(m.astype(np.float32)) > 0)
        want = xc[TC(m)[:, 0]].mean()
        (3.0 * got).backward(); (3.0 * want).backward()
        assert abs(float(got) - float(want)) < 2e-6 * abs(float(want))
        assert rel_err(np_(xg.grad), xc.grad.numpy()) < 2e-6
    x1 = T(synth.normal(42, "mm/flat", (N,))).requires_grad_(True)
    m1 = T((synth.uniform(42, "mm/mflat", (N,)) < 0.5).astype(np.float32))
    got = LU.masked_mean(x1, m1)
    assert abs(float(got) - float((x1 * m1).sum() / m1.sum())) < 1e-6


@pytest.mark.parametrize("N", [2048, 37, 70001])
@pytest.mark.parametrize("keys", ["all", "render_only"])
def test_total_loss_matches_the_boolean_gather_form(N, keys):
    """moda_amd.loss_utils.total_loss (moda_loss_terms: one launch each way) against the reference's assembly as written
    (moda.py:540-705: weight * x[mask].mean() per term, boolean gathers; oracle/torch_ref.py total_loss), value, every term and
    the gradient at every input; with the flags' default weights (feat_wt = 0: a zero-weight term still counts) and others."""
    from moda_amd import loss_utils as LU
    names = {"img_loss_samp": (N, 1), "sil_loss_samp": (N, 1), "frnd_loss_samp": (N,), "frame_cyc_dis": (N,), "vis_loss": ()}
    if keys == "all":
        names.update({"flo_loss_samp": (N, 1), "feat_err": (N, 1), "proj_err": (N, 1)})
    vals = {k: np.asarray(np.abs(synth.normal(71, "tl/" + k, sh)), np.float32) for k, sh in names.items()}
    masks = {"sil_at_samp": (synth.uniform(71, "tl/sil", (N, 1)) > 0.3).astype(np.float32),
             "vis_at_samp": (synth.uniform(71, "tl/vis", (N, 1)) > 0.1).astype(np.float32),
             "sil_at_samp_flo": synth.uniform(71, "tl/flo", (N, 1)) > 0.5}
    for wts in (None, dict(img_wt=1.0, sil_wt=0.1, frnd_wt=0.01, flow_wt=1.0, feat_wt=0.01, proj_wt=0.02, vis_wt=1.0, cyc_wt=0.05)):
        w = dict(LU.LOSS_WEIGHTS)
        w.update(wts or {})
        rc = {k: TC(v).clone().requires_grad_(True) for k, v in vals.items()}
        rc.update({k: TC(v) for k, v in masks.items()})
        tot_c, terms_c = tr.total_loss(rc, w)
        tot_c.backward()
        rg = {k: T(v).requires_grad_(True) for k, v in vals.items()}
        rg.update({k: T(v) for k, v in masks.items()})
        tot_g, terms_g = LU.total_loss(rg, wts)
        tot_g.backward()
        assert abs(float(tot_g) - float(tot_c)) < 2e-6 * abs(float(tot_c))
        assert set(terms_g) == set(terms_c)
        for k in terms_c:
            assert abs(float(terms_g[k]) - float(terms_c[k])) <= 2e-6 * abs(float(terms_c[k])), k
        for k in vals:
            gc = rc[k].grad
            if gc is None or float(gc.abs().max()) == 0:        # zero weight
                assert rg[k].grad is None or float(rg[k].grad.abs().max()) == 0, k
                continue
            assert rel_err(np_(rg[k].grad), gc.numpy()) < 2e-6, k


@pytest.mark.parametrize("F,mean_sq", [(3, False), (2, False), (16, True)])
def test_row_dist_fn_against_torch(F, mean_sq):
    """RowDistFn (moda_row_dist) against the eager forms it replaces -- (a - b).norm(2, -1) (loss_utils.py:200, :216-221) and
    (a - b).pow(2).mean(-1) (rendering.py:573-577) -- value and both gradients; a row with a == b has norm 0 and gradient 0,
    as torch's norm backward."""
    N = 1000
    a = synth.normal(72, "rd/a", (N, 1, F)); b = synth.normal(72, "rd/b", (N, 1, F)); g = synth.normal(72, "rd/g", (N, 1))
    b[5] = a[5]
    ac, bc = TC(a).requires_grad_(True), TC(b).requires_grad_(True)
    oc = (ac - bc).pow(2).mean(-1) if mean_sq else (ac - bc).norm(2, -1)
    (oc * TC(g)).sum().backward()
    ag, bg = T(a).requires_grad_(True), T(b).requires_grad_(True)
    og = A.RowDistFn.apply(ag, bg, mean_sq)
    (og * T(g)).sum().backward()
    assert og.shape == oc.shape
    assert rel_err(np_(og), oc.detach().numpy()) < 1e-6
    assert rel_err(np_(ag.grad), ac.grad.numpy()) < 1e-6 and rel_err(np_(bg.grad), bc.grad.numpy()) < 1e-6
    assert float(ag.grad[5].abs().max()) == 0
    og2 = A.RowDistFn.apply(T(a).requires_grad_(True), T(b), mean_sq)           # constant b: no gradient asked for
    assert torch.equal(og2, og)


@pytest.mark.parametrize("M,N,ld", [(5000, 64, 64), (4096, 16, 16), (70001, 4, 8), (5000, 24, 24), (5000, 63, 64), (300, 64, 64), (9000, 32, 72)])
def test_colsum_forms(M, N, ld):
    """moda_colsum_f32 (out += column sums): the wide-load form (tall, <= 64 columns, 16-byte rows) and the generic one it falls
    back to (columns not a multiple of 4, thread count per row not dividing 256, short matrices), against float64; `out` keeps
    what it held."""
    from moda_amd import _lib as L
    x = synth.normal(73, "cs/x", (M, ld))
    xg = T(x)
    base = synth.normal(73, "cs/b", (N,))
    out = T(base).clone()
    L.call("moda_colsum_f32", L.ptr(xg), M, N, ld, L.ptr(out), L.stream())
    want = base.astype(np.float64) + x[:, :N].astype(np.float64).sum(0)
    assert np.abs(np_(out) - want).max() < 2e-5 * np.abs(x[:, :N]).sum(0).max()


def test_zero_pool_slices_are_fresh_zeros_also_under_graph_replay():
    """autograd._ZeroPool: slices of one pre-zeroed block stand in for ~30 torch.zeros per backward pass.  A slice is exclusive;
    every HIP-graph capture gets a block of its own whose fill is part of that graph (two captures in a row included: the key
    is the runtime's capture id), so an accumulation target is zero again at every replay."""
    pool = A._ZeroPool()
    a = pool.get((5, 3), DEV)
    b = pool.get((7,), DEV)
    assert a.shape == (5, 3) and float(a.abs().sum()) == 0 and a.data_ptr() != b.data_ptr()
    a += 1
    assert float(b.abs().sum()) == 0                                   # exclusive slices
    big = pool.get((pool.CAP // 4,), DEV)                              # larger than a slice may be: its own allocation
    assert big.untyped_storage().data_ptr() != a.untyped_storage().data_ptr()
    graphs, outs = [], []
    for _ in range(2):                                                 # two captures back to back, nothing eager in between
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            t = pool.get((16,), DEV)
            t += 2.0
        graphs.append(g)
        outs.append(t)
    assert outs[0].untyped_storage().data_ptr() != outs[1].untyped_storage().data_ptr()
    assert outs[0].untyped_storage().data_ptr() != a.untyped_storage().data_ptr()
    for _ in range(3):
        graphs[0].replay()
        graphs[1].replay()
    torch.cuda.synchronize()
    assert float(outs[0].max()) == 2.0 and float(outs[1].min()) == 2.0  # re-zeroed by each replay, then += 2
    c = pool.get((4,), DEV)                                            # eager again: not a slice of a graph's block
    assert c.untyped_storage().data_ptr() not in (outs[0].untyped_storage().data_ptr(), outs[1].untyped_storage().data_ptr())


@pytest.mark.parametrize("R,S,shared_code", [(64, 128, False), (33, 37, False), (64, 80, True), (3, 50, False), (40, 96, "vis"), (2, 300, "vis")])
def test_chained_backward_of_the_64_wide_nets_equals_the_per_layer_kernels(R, S, shared_code, monkeypatch):
    """bwd64_chain.hip (the hidden layers' dW / db / masked dX chain of a 64-wide network as one launch, dh kept in LDS between
    the layers) against the per-layer gemm_bf16.hip kernels it replaces (MODA_CHAIN64=0) on the skin network: same bf16 operands,
    same roundings of dh between layers -- what differs is the order of the fp32 sums, so every gradient agrees to a few 1e-5
    (a bf16 tie of dh rounding the other way shows as ~1e-3 of single entries).  Tiles: full (128 rows), ragged, one ray = one
    tile, a shared code row."""
    from gpu_helpers import nerf_from_params
    from helpers import rel_l2
    vis = shared_code == "vis"                   # nerf_vis: no code input, one output, and (below) no gradient asked at xyz
    kw = (dict(D=5, W=64, in_channels_xyz=63, in_channels_dir=0, out_channels=1, raw_feat=True) if vis else
          dict(D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=25, raw_feat=True))
    pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
    p = synth.nerf_params(74, "ch/vis" if vis else "ch/skin", **pk)
    xyz = np.float32(0.3) * synth.normal(74, "ch/xyz", (R, S, 3))
    code = None if vis else synth.normal(74, "ch/code", (1 if shared_code else R, 128))
    gout = synth.normal(74, "ch/g", (R, S, kw["out_channels"]))
    emb = moda_amd.Embedding(3, 10)

    def run(chain):
        monkeypatch.setenv("MODA_CHAIN64", "1" if chain else "0")       # 1: both fused launches (hidden chain + the PE ends)
        m = nerf_from_params(p, **kw).train()
        xg = T(xyz).requires_grad_(not vis)      # (the visibility loss detaches its points: the chain runs without the PE ends)
        cg = None if code is None else T(code).requires_grad_(True)
        moda_amd.set_train_precision("bf16")
        try:
            (m.train_forward(xg, emb, code=cg) * T(gout)).sum().backward()
        finally:
            moda_amd.set_train_precision("fp32")
        out = {} if vis else {"d_xyz": xg.grad, "d_code": cg.grad}
        out.update({pn: pt.grad for pn, pt in m.named_parameters() if pt.grad is not None})
        return out

    a, b = run(True), run(False)
    assert a.keys() == b.keys()
    worst = ("", 0.0)
    for k in a:
        e = rel_l2(np_(a[k]), np_(b[k]))
        worst = max(worst, (k, e), key=lambda t: t[1])
        assert e < 2e-3, (k, e)
    print("chained vs per-layer backward, worst rel-L2:", worst)


@pytest.mark.parametrize("name", ["coarse", "feat"])
@pytest.mark.parametrize("R,S", [(64, 128), (33, 37), (3, 21), (130, 64), (1, 8200), (300, 64)])
def test_fused_hidden_layer_backward_of_the_256_wide_net_equals_the_two_launch_route(R, S, name, monkeypatch):
    """bwd256_fused.hip (dW, db and the masked dX of a W -> W hidden layer, W = 256 or 128, from ONE pass over dZ and the layer's
    input; the ReLU mask is the sign of that input) against gemm_bf16.hip's dW form + dX form (MODA_BWD256=0) on the 8 x 256 network
    with view-direction input and on the 5 x 128 feature network (nerf.py:147-198) in the bf16 training mode: the same bf16 operands and the same k order in dX, so the
    gradient at the points is equal to rounding; the weight gradients differ by the order of their fp32 sums.  Tiles: whole
    64-sample tiles, ragged M, fewer tiles than streams, a stream with many tiles."""
    from test_gpu_parity import _nerf_case
    from helpers import rel_l2
    kw, p, _ = _nerf_case(name, seed=75, tag="b256/")          # coarse: 8 x 256 with view directions; feat: 5 x 128, raw 16-channel output
    from gpu_helpers import nerf_from_params
    xyz = np.float32(0.3) * synth.normal(75, "b256/xyz", (R, S, 3))
    dirs = synth.normal(75, "b256/dir", (R, kw["in_channels_dir"])) if kw["in_channels_dir"] else None
    gout = synth.normal(75, "b256/g", (R, S, 4 if name == "coarse" else 16))
    emb = moda_amd.Embedding(3, 10)

    def run(fused):
        monkeypatch.setenv("MODA_BWD256", "1" if fused else "0")
        m = nerf_from_params(p, **kw).train()
        xg = T(xyz).requires_grad_(True)
        dg = None if dirs is None else T(dirs).requires_grad_(True)
        moda_amd.set_train_precision("bf16")
        try:
            (m.train_forward(xg, emb, dir_src=dg) * T(gout)).sum().backward()
        finally:
            moda_amd.set_train_precision("fp32")
        out = {"d_xyz": xg.grad}
        if dg is not None:
            out["d_dir"] = dg.grad
        out.update({pn: pt.grad for pn, pt in m.named_parameters() if pt.grad is not None})
        return out

    a, b = run(True), run(False)
    assert a.keys() == b.keys() and len(a) == (2 + 2 * 12 if name == "coarse" else 1 + 2 * 8)
    worst = ("", 0.0)
    for k in a:
        assert bool(torch.isfinite(a[k]).all()), k
        e = rel_l2(np_(a[k]), np_(b[k]))
        worst = max(worst, (k, e), key=lambda t: t[1])
        assert e < 2e-3, (k, e)
    print("fused vs two-launch 256-wide hidden layers, worst rel-L2:", worst)


@pytest.mark.parametrize("name,R,S", [("coarse", 520, 128), ("feat", 1030, 64), ("coarse", 2048, 128)])
def test_fused_hidden_layer_backward_repeats_bit_for_bit_in_its_input_gradient(name, R, S):
    """A race screen for bwd256_fused.hip's hand-counted LDS-DMA ring (three / four stages, `vmcnt(N)` + one barrier per tile): the
    same backward 24 times -- the gradient at the points has no atomics on its path and must come out bit-identical every time,
    the weight gradients within the order of their fp32 atomics.  Several tiles per stream (520 x 128 = 1040 tiles on 128 streams,
    1030 x 64 on 256) and cfg4's size."""
    from test_gpu_parity import _nerf_case
    from gpu_helpers import nerf_from_params
    from helpers import rel_l2
    kw, p, _ = _nerf_case(name, seed=76, tag="b256r/")
    m = nerf_from_params(p, **kw).train()
    emb = moda_amd.Embedding(3, 10)
    xyz = T(np.float32(0.3) * synth.normal(76, "b256r/xyz", (R, S, 3)))
    dirs = T(synth.normal(76, "b256r/dir", (R, kw["in_channels_dir"]))) if kw["in_channels_dir"] else None
    gout = T(synth.normal(76, "b256r/g", (R, S, 4 if name == "coarse" else 16)))
    moda_amd.set_train_precision("bf16")
    try:
        first = None
        for it in range(24):
            for q in m.parameters():
                q.grad = None
            xg = xyz.clone().requires_grad_(True)
            (m.train_forward(xg, emb, dir_src=dirs) * gout).sum().backward()
            got = (xg.grad.clone(), [q.grad.clone() for q in m.parameters() if q.grad is not None])
            if first is None:
                first = got
                continue
            assert torch.equal(got[0], first[0]), f"d_xyz changed in repetition {it}"
            for a, b in zip(got[1], first[1]):
                assert rel_l2(np_(a), np_(b)) < 2e-5
    finally:
        moda_amd.set_train_precision("fp32")


@pytest.mark.parametrize("name,M,rows", [("coarse", 4096 + 40, 1), ("feat", 3000, 1), ("skin", 64 * 40, 40)])
@pytest.mark.parametrize("mode", ["bf16x6", "bf16x3"])
def test_split_bf16_backward_sign_maps_equal_the_activation_mask(name, M, rows, mode, monkeypatch):
    """Split-bf16 training modes, backward of NeRF.forward (nerf.py:147-198): the dX launches read the ReLU mask as the 1-bit sign
    map their layer's dW launch left behind (gemm_x3.hip, `MODA_X3_MASK_BITS`) instead of the layer's fp32 activations.  Same
    arithmetic: the input gradient (no atomics on its path) is bit-identical to the activation-mask route, the weight gradients
    agree to their atomics' summation order; ragged M."""
    from test_gpu_parity import _nerf_case
    kw, p, m = _nerf_case(name, seed=17, tag="xbits/")
    m.train()
    emb = moda_amd.Embedding(3, 10, alpha=10.0)
    n_code = kw["in_channels_xyz"] - 63
    xyz = np.float32(0.35) * synth.normal(17, name + "/xyz", (M, 3))
    code = T(synth.normal(17, name + "/code", (rows, n_code))) if n_code else None
    dirs = T(synth.normal(17, name + "/dir", (rows, kw["in_channels_dir"]))) if kw["in_channels_dir"] else None
    g = None
    res = {}
    moda_amd.set_train_precision(mode)
    try:
        for bits in ("1", "0"):
            monkeypatch.setenv("MODA_X3_MASK_BITS", bits)
            x = T(xyz).requires_grad_(True)
            for q in m.parameters():
                q.grad = None
            out = m.train_forward(x, emb, code=code, dir_src=dirs)
            if g is None:
                g = T(synth.normal(17, name + "/g", tuple(out.shape)))
            (out * g).sum().backward()
            res[bits] = (out.detach().clone(), x.grad.clone(), [q.grad.clone() for q in m.parameters() if q.grad is not None])
    finally:
        moda_amd.set_train_precision("fp32")
    assert torch.equal(res["1"][0], res["0"][0])
    assert torch.equal(res["1"][1], res["0"][1]) and float(res["1"][1].abs().max()) > 0
    assert len(res["1"][2]) == len(res["0"][2]) > 10
    for a, b in zip(res["1"][2], res["0"][2]):
        assert rel_err(np_(a), np_(b)) < 2e-5          # (split-K atomics: the summation order differs from run to run)


def test_fan_out_sums_gradients_in_one_launch_and_equals_autograd_accumulation():
    """autograd.FanOutFn / fan_scope (round 6): a tensor that feeds k nodes is handed to each as its own alias; its gradient is
    the k incoming gradients summed by ONE `moda_sum_tensors` launch (argument order) instead of k - 1 `add` launches.  Against
    autograd's own accumulation on the same graph: equal to fp32 round-off of a different summation order; eleven consumers
    (> 8: two launches + the ninth-and-later consumers on the tensor itself); no scope / no grad: the tensor itself."""
    x = T(synth.normal(3, "fan/x", (257, 33, 3))).requires_grad_(True)
    ws = [T(synth.normal(3, f"fan/w{k}", (257, 33, 3))) for k in range(11)]

    def loss(get):
        return sum(((get() * w).sin() * (k + 1)).sum() for k, w in enumerate(ws))
    ref, = torch.autograd.grad(loss(lambda: x), [x])
    calls = []
    orig = A.L.call
    A.L.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        with A.fan_scope():
            assert A.fanned(x) is not x
            got, = torch.autograd.grad(loss(lambda: A.fanned(x)), [x])
    finally:
        A.L.call = orig
    assert calls.count("moda_sum_tensors") == 1                       # 8 aliases (one taken by the assert above) -> one launch
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-6
    assert A.fanned(x) is x                                            # outside a scope
    with A.fan_scope(), torch.no_grad():
        assert A.fanned(x) is x
    fan = A.Fan(x, 11)                                                 # a wide fan: eight operands per launch
    calls.clear()
    A.L.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        got2, = torch.autograd.grad(loss(fan), [x])
    finally:
        A.L.call = orig
    assert calls.count("moda_sum_tensors") == 2 and float((got2 - ref).abs().max() / ref.abs().max()) < 1e-6


def test_affine3_equals_the_eager_expressions_bit_for_bit():
    """moda_affine3 (round 6): the lattice jitter of feat_match and the negatives of visibility_loss as one launch each, with the
    rounding sequence of the eager expressions they replace (loss_utils.py:304-306, :137-138)."""
    from moda_amd import loss_utils as LU
    bound = np.asarray([0.21, 0.17, 0.33], np.float32)
    nz = T(synth.normal(5, "af/nz", (1, 8000, 3)))
    q = LU.feat_grid_query(bound, nz.device, 20, True, {"feat_noise": nz})
    base = LU.feat_grid_query(bound, nz.device, 20, False, None)
    eager = base + nz.reshape(base.shape) * T(bound) * 0.05
    assert torch.equal(q, eager)
    r = T(synth.uniform(5, "af/r", (1, 4099, 3)))
    out = torch.empty_like(r)
    b = T(bound)
    b2, nb = b * 2, -b                                                 # (held: a pointer into a dead temporary reads whatever reuses it)
    A.L.call("moda_affine3", A.L.ptr(r.reshape(-1, 3)), None, A.L.ptr(b2), 1.0, A.L.ptr(nb), 4099, A.L.ptr(out), A.L.stream())
    assert torch.equal(out, r * 2 * b[None, None] - b[None, None])
