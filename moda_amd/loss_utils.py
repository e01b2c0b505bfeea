"""The per-ray loss heads `inference_deform` calls behind compositing, with the reference's names and argument
meaning (nnutils/loss_utils.py): visibility_loss :125-149, compute_pts_exp :165-175, feat_match_loss :176-210,
kp_reproj_loss :212-222, kp_reproj :224-270, feat_match :273-405, and the eikonal regulariser of the canonical density
(nerf_gradient :15-47, compute_gradients_sdf :48-71, eikonal_loss :73-104).  Every arithmetic node is a HIP kernel behind an
autograd Function (moda_amd/autograd.py), so the same code serves the no-grad and the training route.

Random tensors the reference draws inside these functions can be injected through `rng` (dict):
'feat_noise' (1, 20^3, 3) standard normals (loss_utils.py:306), 'vis_neg_rand' (1, N*S, 3) uniforms (:137),
'eik_inds' (1000,) ray indices (:81-84), 's3im_perms' (9, 1024) the permutations of S3IM (:683)."""
import numpy as np
import torch

from . import _lib as L
from . import autograd as A


def compute_pts_exp(pts_prob, pts):
    """loss_utils.py:165-175: pts (..., ndepth, 3), pts_prob (..., ndepth) -> (..., 3) expectation."""
    nd = pts_prob.shape[-1]
    return A.PtsExpFn.apply(pts_prob.reshape(-1, nd), pts.reshape(-1, nd, 3))


def _bound_key(bound):
    """(values, dtype) of the caller's `bound`, as the reference indexes it (bound[0..2])."""
    if torch.is_tensor(bound):
        bound = bound.detach().cpu().numpy()
    b = np.asarray(bound).reshape(-1)[:3]
    if b.dtype.kind != "f":
        b = b.astype(np.float64)
    return b, (tuple(float(v) for v in b), b.dtype.str)


def _query_grid(bound, grid_size):
    """loss_utils.py:290-294: query[i,j,k] = (x_i, y_j, z_k) on linspace(-bound, bound, grid_size), flattened.
    The reference's expression `np.linspace(-bound[c], bound[c], grid_size).astype(np.float32)` is evaluated AS IT STANDS, on
    scalars of the caller's dtype: with a float32 `bound` (MoDA's obj_bound) NumPy >= 2 computes the nodes in float32 arithmetic
    (NEP 50: result_type(float32, float32, python float) is float32), NumPy 1.x in float64 -- a few nodes then differ by one
    float32 ulp, which the 2^9 frequency of the positional encoding turns into 8e-4 of nerf_feat's first-layer weight gradient
    (found by the float64-truth fixtures of round 5; python floats were passed here before).  Mirroring the expression keeps
    this function equal to the reference in whatever environment both run."""
    b, _ = _bound_key(bound)
    ax = [np.linspace(-b[c], b[c], grid_size).astype(np.float32) for c in range(3)]
    g = np.empty((grid_size, grid_size, grid_size, 3), np.float32)
    g[..., 0] = ax[0][:, None, None]
    g[..., 1] = ax[1][None, :, None]
    g[..., 2] = ax[2][None, None, :]
    return g.reshape(-1, 3)


def feat_grid_query(bound, device, grid_size=20, is_training=True, rng=None):
    """The lattice feat_match evaluates nerf_feat on (loss_utils.py:290-306): linspace(-bound, bound, 20)^3, jittered by
    0.05 * bound * randn in training (rng['feat_noise'] injects the draw)."""
    b, key = _bound_key(bound)
    query = L.const_tensor(("feat_grid", key, grid_size), device, lambda: _query_grid(bound, grid_size))   # :290-301
    if is_training:                                                                   # :304-306
        nz = (rng or {}).get('feat_noise')
        nz = torch.randn((1,) + tuple(query.shape), device=device) if nz is None else L.dev(nz)
        scale = L.const_tensor(("bound", key), device, lambda: np.asarray(b, np.float32))
        out = torch.empty_like(query)
        L.call("moda_affine3", L.ptr(L.dev(nz).reshape(query.shape)), L.ptr(query), L.ptr(scale), 0.05, None, query.shape[0],
               L.ptr(out), L.stream())                                                # query + (randn * bound) * 0.05, one launch
        query = out
    return query


def feat_match(nerf_feat, embedding_xyz, feats, bound, grid_size=20, use_corr=True, use_ot=False, is_training=True,
               init_pts=None, rt_entropy=False, rng=None, grid=None):
    """loss_utils.py:273-405: feats (n, 16) pixel features -> (pts_pred (n,3), corr_err).
    grid = (query (G,3), vol (G,16)): the lattice and nerf_feat's output on it, when the caller has already evaluated them
    (render_rays' training route does, in the same network call as the rendered features)."""
    if init_pts is not None:
        return _feat_match_local(nerf_feat, embedding_xyz, feats, bound, grid_size, use_corr, use_ot, init_pts, rt_entropy)
    f = L.dev(feats).reshape(-1, feats.shape[-1])
    dev = f.device
    fn = A.NormalizeFn.apply(f)                                                       # :287
    if grid is not None:
        query, vol = grid
    else:
        query = feat_grid_query(bound, dev, grid_size, is_training, rng)
        train = torch.is_grad_enabled() and (f.requires_grad or any(p.requires_grad for p in nerf_feat.parameters()))
        if train:
            vol = nerf_feat.train_forward(query, embedding_xyz)                       # :311-313
        else:
            vol = nerf_feat.fused(query, n_freq=embedding_xyz.N_freqs, alpha=embedding_xyz.alpha)
    vn = A.NormalizeFn.apply(vol)                                                     # :315
    if use_ot:                                                                        # :338-374
        kappa = torch.full((1,), 1.0 / A.SINKHORN_TEMP, device=dev)
    else:                                                                             # :331-332, :376
        kappa = nerf_feat.beta.abs() + 1e-9
    pts_pred, prob = A.FeatMatchFn.apply(fn, vn, query, kappa, bool(use_ot), bool(use_corr or rt_entropy))   # :389
    corr_err = 0
    if use_corr:                                                                      # :386-391
        tt = A.LinearFn.apply(prob, prob, None, 0)                                    # prob prob^T, (n, n)
        corr_err = (tt - torch.eye(tt.shape[0], device=dev)).norm(2, -1)
    if rt_entropy:                                                                    # :397-402 normalised matching entropy
        match_unc = (-prob * prob.clamp(1e-9, 1 - 1e-9).log()).sum(1)[:, None] / float(np.log(grid_size ** 3))
        return pts_pred, match_unc, corr_err
    return pts_pred, corr_err


def _feat_match_local(nerf_feat, embedding_xyz, feats, bound, grid_size, use_corr, use_ot, init_pts, rt_entropy):
    """feat_match with `init_pts` (loss_utils.py:297-300, 322-331): every pixel n matches against a lattice of its own,
    query + init_pts[n] -- n * grid_size^3 network evaluations (one fused launch per chunk of pixels), the per-pixel matrix
    exp((<vol[n, g], feats[n]> - 1) kappa) (moda_match_matrix_rows) and the SAME library tail as the shared-lattice form
    (moda_match_sweep x 40, moda_match_expect, moda_match_prob; round 5 -- it was torch arithmetic before).  No caller in the
    reference uses this option (scripts/visualize/match.py:101, loss_utils.py:197): inference only, no lattice jitter (:304)."""
    L.no_grad_only(feats, init_pts, *nerf_feat.parameters())
    f = L.dev(feats).reshape(-1, feats.shape[-1])
    dev = f.device
    fn = A.NormalizeFn.apply(f)                                                       # :287
    base = L.const_tensor(("feat_grid", _bound_key(bound)[1], grid_size), dev, lambda: _query_grid(bound, grid_size))
    init = L.dev(init_pts).reshape(-1, 3)
    query = base[None] + init[:, None]                                                # :298 (n, G, 3)
    n, G = query.shape[0], query.shape[1]
    use_ot = bool(use_ot)
    kappa = torch.full((1,), 1.0 / A.SINKHORN_TEMP, device=dev) if use_ot else (nerf_feat.beta.abs() + 1e-9).reshape(1)   # :340 / :331-332
    Kmat = torch.empty((n, G), device=dev)
    step = max(1, (1 << 22) // G)                                                     # pixels per network launch (~4 M points)
    for j in range(0, n, step):
        vol = nerf_feat.fused(query[j:j + step].contiguous(), n_freq=embedding_xyz.N_freqs, alpha=embedding_xyz.alpha)   # :325-326
        vn = A.NormalizeFn.apply(vol.reshape(-1, vol.shape[-1]))                      # :328
        m = min(step, n - j)
        # K[n, g] = exp((<vol[n, g], feats[n]> - 1) kappa): the Sinkhorn kernel of :340 / the shifted softmax numerator of :337, 376
        L.call("moda_match_matrix_rows", L.ptr(fn[j:j + m]), L.ptr(vn), m, G, f.shape[1], L.ptr(kappa), L.ptr(Kmat[j:j + m]), L.stream())
    b = None
    if use_ot:                                                                        # :338-374: the sweeps of the shared-lattice form
        KmatT = Kmat.t().contiguous()
        a_t = torch.full((n,), 1.0 / n, device=dev)
        b = torch.empty((G,), device=dev)
        for _ in range(A.SINKHORN_ITERS):
            L.call("moda_match_sweep", L.ptr(KmatT), G, n, L.ptr(a_t), 1, 1.0 / G, None, L.ptr(b), 0, L.stream())
            a_n = torch.empty_like(a_t)
            L.call("moda_match_sweep", L.ptr(Kmat), n, G, L.ptr(b), 1, 1.0 / n, None, L.ptr(a_n), 0, L.stream())
            a_t = a_n
    # expectation over the shared base lattice, then the pixel's own offset: sum_g prob (base_g + init_n) = E[base] + init_n (:395)
    pred = torch.empty((n, 3), device=dev)
    rowsum = torch.empty((n,), device=dev)
    L.call("moda_match_expect", L.ptr(Kmat), L.ptr(b), L.ptr(base), n, G, L.ptr(pred), L.ptr(rowsum), 0, L.stream())
    pts_pred = pred + init
    corr_err = 0
    prob = None
    if use_corr or rt_entropy:
        prob = torch.empty((n, G), device=dev)
        L.call("moda_match_prob", L.ptr(Kmat), L.ptr(b), L.ptr(rowsum), n, G, L.ptr(prob), 0, L.stream())
    if use_corr:                                                                      # :386-391
        tt = A.LinearFn.apply(prob, prob, None, 0)                                    # prob prob^T (library GEMM)
        corr_err = (tt - torch.eye(n, device=dev)).norm(2, -1)
    if rt_entropy:
        match_unc = (-prob * prob.clamp(1e-9, 1 - 1e-9).log()).sum(1)[:, None] / float(np.log(grid_size ** 3))
        return pts_pred, match_unc, corr_err
    return pts_pred, corr_err


def feat_match_loss(nerf_feat, embedding_xyz, feats, pts, pts_prob, bound, use_corr=True, use_ot=False,
                    is_training=True, rng=None, grid=None):
    """loss_utils.py:176-210 -> (pts_pred (...,3), pts_exp (...,3), feat_err (...,1), corr_err)."""
    base = tuple(feats.shape[:-1])
    pts_exp = compute_pts_exp(pts_prob, pts)                                          # :193
    pts_pred, corr_err = feat_match(nerf_feat, embedding_xyz, feats, bound, grid_size=20, use_corr=use_corr,
                                    use_ot=use_ot, is_training=is_training, rng=rng, grid=grid)  # :196-197
    feat_err = A.RowDistFn.apply(pts_pred, pts_exp)                                   # :200 (pts_pred - pts_exp).norm(2, -1)
    if use_corr:
        corr_err = corr_err.view(base + (1,))                                         # :207-208
    return pts_pred.view(base + (3,)), pts_exp.view(base + (3,)), feat_err.view(base + (1,)), corr_err


def forward_warp(pts, models, embedding_xyz, bone_rts, dskin=None, dskin_bns=False, pts_tf=None):
    """Canonical -> observed warp of pts (N,n,3) with the rest-pose skinning field (gauss_mlp_skinning with
    rest_pose_code + neu_dbs backward=False; rendering.py:351-352, loss_utils.py:250-254).
    `dskin`: nerf_skin's output at exactly these points with the rest-pose code, when the caller already has it (the
    reference re-evaluates the same network on the same inputs for the cycle, target and dense-target warps,
    rendering.py:330, :351, :356); (N,n,B), or (N,B,n) with dskin_bns.  `pts_tf`: the points the transform is applied to
    when a residual field displaces them first (pts + nerf_dis(pts, rest), geom_utils.py:420-425)."""
    bones_rst = L.dev(models['bones_rst'])
    B = bones_rst.shape[-2]
    N, n_s = pts.shape[0], pts.shape[1]
    nerf_skin = models['nerf_skin'] if 'nerf_skin' in models.keys() else None
    rest = models['rest_pose_code'].weight.reshape(1, -1)
    leaves = [pts, bones_rst, bone_rts, models['skin_aux'], rest] + ([] if nerf_skin is None else list(nerf_skin.parameters()))
    if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in leaves + [dskin, pts_tf]):
        ds = dskin
        if ds is None and nerf_skin is not None:
            ds = nerf_skin.train_forward(pts, embedding_xyz, code=A.fanned(models['rest_pose_code'].weight).reshape(1, -1))
        rts = L.dev(bone_rts).reshape(-1, B, 8)
        if rts.shape[0] == 0 or N % rts.shape[0]:
            raise ValueError(f"bone_rts: {rts.shape[0]} transform sets do not divide {N} rays")
        if rts.shape[0] != N:                          # per-frame rows under autograd: expanded (gradients sum per frame)
            rts = A.ExpandRowsFn.apply(rts.reshape(rts.shape[0], B * 8), N // rts.shape[0]).reshape(N, B, 8)
        return A.WarpFn.apply(A.bone_prep(A.fanned(bones_rst).reshape(1, B, 10)), rts, pts, ds,
                              A.fanned(L.dev(models['skin_aux'])), None, pts_tf)[0]
    from .geom_utils import warp                      # no graph wanted: the fused inference kernels
    ds, bns = dskin, dskin_bns
    if ds is None and nerf_skin is not None:
        ds = nerf_skin.fused(pts, n_freq=embedding_xyz.N_freqs, alpha=embedding_xyz.alpha, code=L.dev(rest), out_tr_S=n_s)
        bns = True
    # frame-grouped layout (rays['rays_per_frame']): bone_rts may hold one row per frame of k consecutive rays
    n_sets = L.dev(bone_rts).reshape(-1, B * 8).shape[0]
    if n_sets == 0 or N % n_sets:
        raise ValueError(f"bone_rts: {n_sets} transform sets do not divide {N} rays")
    return warp(bones_rst, bone_rts, pts, ds, models['skin_aux'], backward=False, dskin_bns=bns, pts_tf=pts_tf,
                rays_per_set=N // n_sets)[0]


def kp_reproj(pts_pred, models, embedding_xyz, rays, to_target=False, neudbs=True):
    """loss_utils.py:224-270: canonical points (...,3) -> pixel coordinates (N,1,2) in the (target) frame."""
    if not neudbs:
        raise NotImplementedError("linear blend skinning: MoDA runs neudbs (moda.py:72-73)")
    pts = pts_pred.reshape(-1, 1, 3)
    N = pts.shape[0]
    rtk = L.dev(rays['rtk_vec_target'] if to_target else rays['rtk_vec']).reshape(N, 21)
    if 'bones' in models.keys():
        pts = forward_warp(pts, models, embedding_xyz, rays['bone_rts_target'] if to_target else rays['bone_rts'])
    return A.ProjectFn.apply(pts, rtk)[..., :2]


def kp_reproj_loss(pts_pred, xys, models, embedding_xyz, rays, neudbs=True):
    """loss_utils.py:212-222 -> reprojection distance (...,1)."""
    xy = kp_reproj(pts_pred, models, embedding_xyz, rays, neudbs=neudbs)
    err = A.RowDistFn.apply(xy, L.dev(xys).reshape(-1, 1, 2))            # (xys - xy).norm(2, -1)
    return err.view(tuple(pts_pred.shape[:-1]) + (1,))


def visibility_loss(mlp, embed, xyz_pos, w_pos, bound, chunk, rng=None):
    """loss_utils.py:125-149: xyz_pos (N,S,3) with visibilities w_pos (N,S); negatives uniform in the bound."""
    xyz_pos = L.dev(xyz_pos).detach()
    w_pos = L.dev(w_pos).detach()
    dev = xyz_pos.device
    nsample = w_pos.shape[0] * w_pos.shape[1]
    bt = tuple(float(b) for b in np.asarray(bound).reshape(-1)[:3])
    r = (rng or {}).get('vis_neg_rand')
    r = torch.rand(1, nsample, 3) if r is None else r                                 # :137 (the reference draws on the CPU)
    bnd2 = L.const_tensor(("bound*2", bt), dev, lambda: np.asarray(bt, np.float32) * np.float32(2))
    nbnd = L.const_tensor(("-bound", bt), dev, lambda: -np.asarray(bt, np.float32))
    xyz_neg = torch.empty((1, nsample, 3), device=dev)
    L.call("moda_affine3", L.ptr(L.dev(r.to(dev)).reshape(nsample, 3)), None, L.ptr(bnd2), 1.0, L.ptr(nbnd), nsample,
           L.ptr(xyz_neg), L.stream())                                                # (rand * 2) * bound - bound (:138), one launch
    train = torch.is_grad_enabled() and any(p.requires_grad for p in mlp.parameters())

    def logits(x):
        if train:
            return mlp.train_forward(x, embed)[..., 0]
        return mlp.fused(x, n_freq=embed.N_freqs, alpha=embed.alpha, with_sigma=False, sigmoid=False)[..., 0]

    # negatives and positives through ONE evaluation of the network (the reference makes two, :139 and :144; same arithmetic
    # per point): one forward / backward launch chain and one set of weight-gradient GEMMs instead of two
    both = logits(torch.cat([xyz_neg.reshape(-1, 3), xyz_pos.reshape(-1, 3)], 0))
    return A.VisPairLossFn.apply(both, w_pos, nsample)                                 # :140 + :145


def masked_mean(x, mask):
    """`x[mask].mean()` as the reference's loss assembly writes it (moda.py:540-640), as one kernel each way and without the
    boolean gather's host sync: x (N, k) | (N,), mask (N, 1) | (N,) bool or float (non-zero = selected).  NaN when nothing is
    selected, as the reference's mean of an empty selection."""
    return A.MaskedMeanFn.apply(x, mask)


# the weights of moda.py's loss assembly (flags moda.py:153-162; the visibility term's 0.01 is written out at :702)
LOSS_WEIGHTS = dict(img_wt=0.1, sil_wt=0.1, frnd_wt=1.0, flow_wt=1.0, feat_wt=0.0, proj_wt=0.02, cyc_wt=1.0, vis_wt=0.01)
LOSS_TERMS = ("img", "sil", "frnd", "flo", "feat", "proj", "vis", "cyc")


def total_loss(rendered, weights=None):
    """The total loss of banmo.forward_default over render_rays' result dict (moda.py:540-705), default configuration (no
    loss_flt / rm_novp / warm-up branches): img_wt * img_loss_samp[sil > 0].mean() + sil_wt * sil_loss_samp[vis > 0].mean() +
    frnd_wt * frnd_loss_samp[sil > 0].mean() + 2 flow_wt * flo_loss_samp[sil_at_samp_flo].mean() + feat_wt * feat_err[sil > 0]
    .mean() + proj_wt * proj_err[sil > 0].mean() + cyc_wt * frame_cyc_dis.mean() + vis_wt * vis_loss -- the terms whose keys
    the dict holds, as ONE launch forward and one backward (moda_loss_terms) instead of ~55 eager ops with a boolean gather
    (and its host sync) per term.  -> (total, {term name: weighted term, detached}).  moda.py itself cannot be imported here
    (absl, mcubes ...): restated from the cited lines, checked against the plain-torch restatement (oracle/torch_ref.py)."""
    w = dict(LOSS_WEIGHTS)
    w.update(weights or {})
    sil = rendered.get("sil_at_samp")
    plan = [("img", "img_loss_samp", w["img_wt"], sil, ">0"), ("sil", "sil_loss_samp", w["sil_wt"], rendered.get("vis_at_samp"), ">0"),
            ("frnd", "frnd_loss_samp", w["frnd_wt"], sil, ">0"),
            ("flo", "flo_loss_samp", 2.0 * w["flow_wt"], rendered.get("sil_at_samp_flo"), "bool"),
            ("feat", "feat_err", w["feat_wt"], sil, ">0"), ("proj", "proj_err", w["proj_wt"], sil, ">0"),
            ("vis", "vis_loss", w["vis_wt"], None, None), ("cyc", "frame_cyc_dis", w["cyc_wt"], None, None)]
    names, spec, xs = [], [], []
    for name, key, wt, mask, kind in plan:
        if key in rendered:
            names.append(name)
            spec.append((wt, mask, kind))
            xs.append(rendered[key])
    total, terms = A.LossTermsFn.apply(spec, *xs)
    return total, {n: terms[i] for i, n in enumerate(names)}


def unc_loss(rendered):
    """The uncertainty network's own loss (moda.py:707-720): mean over rays of (sil_at_samp * img_loss_samp.mean(-1)).detach() -
    unc_pred[..., 0], squared.  Only nerf_unc receives a gradient from it."""
    target = (L.dev(rendered["sil_at_samp"])[..., 0] * rendered["img_loss_samp"].mean(-1)).detach()
    return A.RowDistFn.apply(rendered["unc_pred"].reshape(-1, 1), target.reshape(-1, 1), True).mean()


def s3im_loss(src_vec, tar_vec, mask, kernel_size=4, stride=4, repeat_time=10, patch_height=32, patch_width=32, rng=None):
    """S3IM(kernel_size, stride, repeat_time, patch_height, patch_width)(src_vec, tar_vec, mask) of loss_utils.py:648-702 with
    the constructor arguments rendering.py:529 passes as defaults: 1 - SSIM(window 4, stride 4) between the rendered and
    the observed colours (both times the mask) re-arranged into a (1, 3, 32, 32 * 10) virtual patch -- the first 1024 rays
    (rows repeated when there are fewer), once in order and nine times permuted (torch.randperm on the CPU, as there; or
    rng['s3im_perms'], (repeat_time - 1, 1024) integers).  One kernel forward, one backward (csrc/loss_kernels.hip).
    Unlike the reference this does NOT multiply its arguments by the mask in place; render_rays mirrors that side effect."""
    if kernel_size != 4 or stride != 4:
        raise NotImplementedError("the S3IM kernel implements the reference's call: kernel_size = stride = 4 (rendering.py:529)")
    src = L.dev(src_vec).reshape(-1, 3)
    tar = L.dev(tar_vec).reshape(-1, 3)
    P = patch_height * patch_width
    perms = (rng or {}).get('s3im_perms')
    if perms is None:
        perms = torch.stack([torch.randperm(P) for _ in range(repeat_time - 1)]) if repeat_time > 1 else torch.zeros((0, P))
    perms = torch.as_tensor(perms).to(device=src.device, dtype=torch.int32).reshape(-1)
    if perms.numel() != (repeat_time - 1) * P:
        raise ValueError(f"s3im_perms: expected {(repeat_time - 1, P)} indices, got {perms.numel()}")
    ident = L.const_tensor(("arange", P), src.device, lambda: torch.arange(P)).to(torch.int32)
    index = torch.cat([ident, perms])                                                  # :678-686
    return A.S3imFn.apply(src, tar, L.dev(mask).reshape(-1, 1).expand(src.shape[0], 1), index, patch_height)


def nerf_gradient(mlp, embed, pts, use_xyz=False, code=None, sigma_only=False):
    """loss_utils.py:15-47 for the density head: (d y / d pts (..., 3, 1), sigmas (..., 1)), the gradient differentiable
    with respect to the parameters.

    The reference differentiates autograd's own backward pass (create_graph=True).  Here the reverse sweep is written
    out as forward nodes -- g <- (g * relu_mask_i) W_i from the density row back to the encoding, then the encoding's
    Jacobian -- so one ordinary backward gives the same parameter gradient (ReLU'' = 0: the masks are constants, which
    is also what the double backward sees)."""
    if not sigma_only or code is not None or use_xyz:
        raise NotImplementedError("nerf_gradient serves eikonal_loss: sigma_only=True, no code (loss_utils.py:97)")
    if mlp.skips != [4]:
        raise NotImplementedError("skips=[4] (the only value MoDA uses)")
    lead = pts.shape[:-1]
    x = L.dev(pts).detach().reshape(-1, 3)
    P = x.shape[0]
    with torch.no_grad():                                   # primal pass: activation signs and y
        emb = embed(x)
        h, masks = emb, []
        for i in range(mlp.D):
            if i in mlp.skips:
                h = torch.cat([emb, h], -1)
            h = mlp._linear(h, getattr(mlp, f"xyz_encoding_{i+1}")[0], 1)
            masks.append(h > 0)
        y = mlp._linear(h, mlp.sigma, 0)
        sdf = -y
        sigmas = 0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / (mlp.beta.abs() + 1e-9))
    ne = emb.shape[1]
    g = mlp.sigma.weight.expand(P, mlp.W)
    g_emb = None
    for i in reversed(range(mlp.D)):
        lin = getattr(mlp, f"xyz_encoding_{i+1}")[0]
        g = A.LinearFn.apply(g * masks[i], lin.weight.t(), None, 0)          # (P, fan_in) = (g . mask) W
        if i in mlp.skips:
            g_emb, g = g[:, :ne], g[:, ne:]
    g_emb = g if g_emb is None else g_emb + g
    # encoding Jacobian (nerf.py:58-72 order: x, then sin(2^k x), cos(2^k x) per band, each times the window w_k): one kernel,
    # differentiable w.r.t. g_emb through the encoding's tangent kernel (A.EmbedJacTFn; torch.sin / cos loops before round 5)
    out = A.EmbedJacTFn.apply(x, g_emb.contiguous(), embed.N_freqs, [float(w) for w in embed.window()])
    return out.view(lead + (3, 1)), sigmas.view(lead + (1,))


def compute_gradients_sdf(mlp, embed, pts, sigma_only=False, eps=1e-3):
    """loss_utils.py:48-71: tetrahedral finite differences, four density evaluations (whole-network nodes)."""
    pts = L.dev(pts).detach()
    ks = ((1, -1, -1), (-1, -1, 1), (-1, 1, -1), (1, 1, 1))
    total = None
    for kk in ks:
        k = L.const_tensor(("tetra", kk), pts.device, lambda: np.asarray(kk, np.float32))
        sdf = mlp.train_forward(pts + k * eps, embed, sigma_only=sigma_only)
        total = k * sdf if total is None else total + k * sdf
    return total / (4.0 * eps)


def eikonal_loss(mlp, embed, pts, bound, ppr_eikonal, rng=None):
    """loss_utils.py:73-104: mean (|d sigma / d x| - 1)^2 over (at most 1000 rays of) the in-bound canonical points.
    Out-of-bound points are weighted 0 instead of being gathered out, so no size is read back from the device."""
    bs = pts.shape[0]
    if bs > 1000:                                                                                     # :79-84
        ri = (rng or {}).get('eik_inds')
        ri = torch.multinomial(torch.ones(bs), 1000, replacement=False) if ri is None else ri
        pts = pts[ri.to(pts.device)]
    pts = L.dev(pts).detach().reshape(-1, 3)
    bt = tuple(float(b) for b in np.asarray(bound).reshape(-1)[:3])
    bnd = L.const_tensor(("bound", bt), pts.device, lambda: np.asarray(bt, np.float32))[None]
    inb = (((bnd - pts.abs()) > 0).sum(-1) == 3).float()                                              # :91
    if ppr_eikonal:
        g = compute_gradients_sdf(mlp, embed, pts, sigma_only=True)
    else:
        g = nerf_gradient(mlp, embed, pts, sigma_only=True)[0][..., 0]
    err = (g.norm(2, dim=-1) - 1) ** 2
    return (err * inb).sum() / inb.sum()
