"""CPU oracle for MoDA's per-ray rendering hot path -- TEST INFRASTRUCTURE ONLY.

A numpy restatement of the reference algorithm (ChaoyueSong/MoDA,
nnutils/rendering.py + nnutils/nerf.py + nnutils/dual_quat.py + the skinning
subset of nnutils/geom_utils.py).  Every function cites the reference
file:line it follows.  It is the *checker*: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The product package
(moda_amd/) never does; its compute runs in the HIP library only.

Parity pin: checked in this repo's CPU test-suite against golden vectors that
were produced by importing the reference itself in the development container
(tests/golden/gen_golden.py -> tests/golden/*.npz).  The reference ships no
tests of its own for this path, and its one third-party
arithmetic dependency, pytorch3d.transforms, is absent and version-unpinned
(reference .gitmodules:7-9), so the four quaternion helpers below restate the
published closed forms -- for those four functions parity is pinned only
through the reference's call sites (bone_transform, vec_to_sim3).

All functions are dtype-generic: they compute in the dtype of their inputs
(float32 to mirror the reference, float64 for a high-precision truth).
Random tensors the reference draws internally (rendering.py:82,193,389,607)
are explicit arguments here.
"""
import numpy as np

# ----------------------------------------------------------------------------
# pytorch3d.transforms closed forms (absent third-party dependency)
# ----------------------------------------------------------------------------


def quaternion_raw_multiply(a, b):
    """Real-first Hamilton product a (x) b (pytorch3d.transforms.quaternion_raw_multiply)."""
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    ow = aw * bw - ax * bx - ay * by - az * bz
    ox = aw * bx + ax * bw + ay * bz - az * by
    oy = aw * by - ax * bz + ay * bw + az * bx
    oz = aw * bz + ax * by - ay * bx + az * bw
    return np.stack((ow, ox, oy, oz), -1)


def quaternion_multiply(a, b):
    """Product with the real part made non-negative (pytorch3d quaternion_multiply)."""
    ab = quaternion_raw_multiply(a, b)
    return np.where(ab[..., 0:1] < 0, -ab, ab)


def quaternion_invert(q):
    return q * np.asarray([1, -1, -1, -1], dtype=q.dtype)


def quaternion_to_matrix(q):
    """Rotation matrix of a (not necessarily unit) real-first quaternion, scaled by 2/|q|^2."""
    r, i, j, k = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    two_s = q.dtype.type(2.0) / (q * q).sum(-1)
    one = q.dtype.type(1.0)
    o = np.stack(
        (
            one - two_s * (j * j + k * k),
            two_s * (i * j - k * r),
            two_s * (i * k + j * r),
            two_s * (i * j + k * r),
            one - two_s * (i * i + k * k),
            two_s * (j * k - i * r),
            two_s * (i * k - j * r),
            two_s * (j * k + i * r),
            one - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


# ----------------------------------------------------------------------------
# nnutils/dual_quat.py
# ----------------------------------------------------------------------------


def q_normalize(q):
    """dual_quat.py:4-12 (input is (n,4): the reference indexes norm[:, None])."""
    assert q.shape[-1] == 4
    norm = np.sqrt(np.sum(np.square(q), axis=-1))
    assert not np.any(np.isclose(norm, np.zeros_like(norm)))
    return q / norm[:, None]


def q_mul(q1, q2):
    """dual_quat.py:14-31: outer product q2 q1^T and signed sums == Hamilton q1 (x) q2."""
    assert q1.shape[-1] == 4 and q2.shape[-1] == 4
    shape = q1.shape
    a = q1.reshape(-1, 4)
    b = q2.reshape(-1, 4)
    t = b[:, :, None] * a[:, None, :]  # terms[i][j] = q2[i] * q1[j]
    w = t[:, 0, 0] - t[:, 1, 1] - t[:, 2, 2] - t[:, 3, 3]
    x = t[:, 0, 1] + t[:, 1, 0] - t[:, 2, 3] + t[:, 3, 2]
    y = t[:, 0, 2] + t[:, 1, 3] + t[:, 2, 0] - t[:, 3, 1]
    z = t[:, 0, 3] - t[:, 1, 2] + t[:, 2, 1] + t[:, 3, 0]
    return np.stack((w, x, y, z), 1).reshape(shape)


def dq_mul(dq1, dq2):
    """dual_quat.py:33-49."""
    assert dq1.shape[-1] == 8 and dq2.shape[-1] == 8
    r1, d1 = dq1[..., :4], dq1[..., 4:]
    r2, d2 = dq2[..., :4], dq2[..., 4:]
    return np.concatenate([q_mul(r1, r2), q_mul(r1, d2) + q_mul(d1, r2)], -1)


def dq_normalize(dq):
    """dual_quat.py:51-62: divide all 8 coefficients by the real part's norm."""
    assert dq.shape[-1] == 8
    norm = np.sqrt(np.sum(np.square(dq[..., :4]), axis=-1, keepdims=True))
    assert not np.any(np.isclose(norm, np.zeros_like(norm)))
    return dq / norm


def dq_quaternion_conjugate(dq):
    """dual_quat.py:65-74."""
    assert dq.shape[-1] == 8
    return dq * np.asarray([1, -1, -1, -1, 1, -1, -1, -1], dtype=dq.dtype)


def dq_combined_conjugate(dq):
    """dual_quat.py:76-85."""
    assert dq.shape[-1] == 8
    return dq * np.asarray([1, -1, -1, -1, -1, 1, 1, 1], dtype=dq.dtype)


def dq_inverse(dq):
    """dual_quat.py:87-94: conj_q(dq) / |dq_r|^2."""
    return dq_quaternion_conjugate(dq) / np.sum(np.square(dq[..., :4]), axis=-1)[..., None]


# ----------------------------------------------------------------------------
# nnutils/nerf.py
# ----------------------------------------------------------------------------


def embedding_window(n_freqs, alpha, dtype):
    """nerf.py:63-68: w_k = 0.5 (1 + cos(pi clamp(alpha-k,0,1) + pi))."""
    win = np.clip(dtype.type(alpha) - np.arange(n_freqs).astype(dtype), 0.0, 1.0)
    return (0.5 * (1 + np.cos(dtype.type(np.pi) * win + dtype.type(np.pi)))).astype(dtype)


def embedding(x, n_freqs, alpha=None):
    """nerf.py:35-75: [x, w_k sin(2^k x), w_k cos(2^k x)]_k, freq-major then (sin,cos) then channel."""
    if n_freqs <= 0:
        return x
    if alpha is None:
        alpha = n_freqs
    dt = x.dtype
    shape = x.shape
    c = shape[-1]
    xf = x.reshape(-1, c)
    win = embedding_window(n_freqs, alpha, dt)
    out = [xf]
    for k in range(n_freqs):
        f = dt.type(2.0**k)
        out.append(win[k] * np.sin(f * xf))
        out.append(win[k] * np.cos(f * xf))
    return np.concatenate(out, -1).reshape(shape[:-1] + (c * (1 + 2 * n_freqs),))


def _linear(x, w, b):
    return x @ w.T + b


def _relu(x):
    return np.maximum(x, 0)


def _sigmoid(x):
    return 1 / (1 + np.exp(-x))


def nerf_forward(p, x, D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=(4,),
                 raw_feat=False, sigma_only=False, round_fn=None):
    """nerf.py:147-198.  `p` maps the reference's state-dict names to arrays.

    round_fn, when given, is applied to the weights and to every layer input
    before each product (used to emulate the product's bf16 MFMA mode: operands
    rounded to bf16, fp32 accumulate).  The reference itself is round_fn=None.
    """
    r = (lambda a: a) if round_fn is None else round_fn
    if not sigma_only:
        input_xyz, input_dir = x[..., :in_channels_xyz], x[..., in_channels_xyz:in_channels_xyz + in_channels_dir]
    else:
        input_xyz, input_dir = x[..., :in_channels_xyz], x[..., :0]
    h = input_xyz
    for i in range(D):
        if i in skips:
            h = np.concatenate([input_xyz, h], -1)
        h = _relu(_linear(r(h), r(p[f"xyz_encoding_{i+1}.0.weight"]), p[f"xyz_encoding_{i+1}.0.bias"]))
    sigma = _linear(r(h), r(p["sigma.weight"]), p["sigma.bias"])
    if sigma_only:
        return sigma
    final = _linear(r(h), r(p["xyz_encoding_final.weight"]), p["xyz_encoding_final.bias"])
    d = _relu(_linear(r(np.concatenate([final, input_dir], -1)), r(p["dir_encoding.0.weight"]), p["dir_encoding.0.bias"]))
    rgb = _linear(r(d), r(p["rgb.0.weight"]), p["rgb.0.bias"])
    if raw_feat:
        return rgb
    return np.concatenate([_sigmoid(rgb), sigma], -1)


def bf16_round(a):
    """Round-to-nearest-even float32 -> bfloat16 -> float32 (what v_cvt_pk_bf16_f32 does)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32)
    rounded = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000))
    out = rounded.view(np.float32).copy()
    nan = np.isnan(a)
    out[nan] = a[nan]
    return out


def f16_round(a):
    """Round-to-nearest-even float32 -> float16 -> float32 (what v_cvt_pk_f16_f32 does; |a| >= 65520 becomes inf): the
    `round_fn` hook that emulates the fp16 mode of the fused kernels."""
    with np.errstate(over="ignore"):
        return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


# ----------------------------------------------------------------------------
# nnutils/geom_utils.py (skinning subset)
# ----------------------------------------------------------------------------


def evaluate_mlp(model_fn, xyz_embedded, embed_fn=None, dir_embedded=None, chunk=32 * 1024,
                 code=None, appearance_code=None, sigma_only=False):
    """geom_utils.py:19-57.  model_fn(x, sigma_only=...) evaluates a NeRF on (n,nbins,k) input."""
    B, nbins, _ = xyz_embedded.shape
    outs = []
    if code is not None and code.shape[0] != B and code.ndim == 2:
        code = np.repeat(code, B, 0)  # (1,c) -> (B,c)  (geom_utils.py:38-39)
    for i in range(0, B, chunk):
        emb = xyz_embedded[i:i + chunk]
        if embed_fn is not None:
            emb = embed_fn(emb)
        if dir_embedded is not None:
            emb = np.concatenate([emb, dir_embedded[i:i + chunk]], -1)
        if code is not None:
            cc = code[i:i + chunk]
            if cc.ndim == 2:
                cc = cc[:, None]
            emb = np.concatenate([emb, np.repeat(cc, nbins, 1)], -1)
        if appearance_code is not None:
            ac = appearance_code[i:i + chunk]
            if ac.ndim == 2:
                ac = ac[:, None]
            emb = np.concatenate([emb, np.repeat(ac, nbins, 1)], -1)
        outs.append(model_fn(emb, sigma_only=sigma_only))
    return np.concatenate(outs, 0)


def bone_transform(bones_in, rts):
    """geom_utils.py:59-111, neudbs branch: bones (B,10), rts (...,B*8) -> (bs,B,10)."""
    B = bones_in.shape[-2]
    bones = bones_in.reshape(-1, B, 10)
    center, orient, scale = bones[:, :, :3], bones[:, :, 3:7], bones[:, :, 7:10]
    rts = rts.reshape(-1, B, 8)
    bs = rts.shape[0]
    dq_r, dq_d = rts[..., :4], rts[..., 4:]
    Rmat = quaternion_to_matrix(dq_r)
    Tmat = 2 * quaternion_raw_multiply(dq_d, quaternion_invert(dq_r))[..., 1:]
    center = (Rmat @ np.broadcast_to(center, (bs, B, 3))[..., None])[..., 0] + Tmat
    orient = quaternion_multiply(dq_r, np.broadcast_to(orient, (bs, B, 4)))
    scale = np.broadcast_to(scale, (bs, B, 3)) if scale.shape[0] == 1 else scale
    return np.concatenate([center, orient, scale], -1).astype(bones_in.dtype)


def vec_to_sim3(vec):
    """geom_utils.py:187-199."""
    center = vec[..., :3]
    orient = vec[..., 3:7]
    nrm = np.maximum(np.sqrt((orient * orient).sum(-1, keepdims=True)), vec.dtype.type(1e-12))
    orient = quaternion_to_matrix(orient / nrm)
    scale = np.exp(vec[..., 7:10])
    return center, orient, scale


def skinning(bones, pts, dskin=None, skin_aux=None, chunk=1024):
    """geom_utils.py:237-302: Gaussian (Mahalanobis) bone logits + dskin, softmax over bones."""
    if pts.shape[0] > chunk:  # ray chunks (geom_utils.py:293-300; chunk size does not change results)
        outs = []
        for i in range(0, pts.shape[0], chunk):
            outs.append(skinning(bones if bones.ndim == 2 else bones[i:i + chunk], pts[i:i + chunk],
                                 None if dskin is None else dskin[i:i + chunk], skin_aux, chunk))
        return np.concatenate(outs, 0)
    bs, N, _ = pts.shape
    B = bones.shape[-2]
    if bones.ndim == 2:
        bones = np.broadcast_to(bones[None], (bs, B, 10))
    bones = bones.reshape(-1, B, 10)
    log_scale = skin_aux[0]
    center, orient, scale = vec_to_sim3(bones)
    orient_t = np.swapaxes(orient, -1, -2)  # R^T (geom_utils.py:252)
    mdis = center.reshape(bs, 1, B, 3) - pts.reshape(bs, N, 1, 3)
    mdis = (orient_t.reshape(bs, 1, B, 3, 3) * mdis.reshape(bs, N, B, 1, 3)).sum(4)
    mdis = scale.reshape(bs, 1, B, 3) * mdis**2
    mdis = mdis * 100 * np.exp(log_scale)
    mdis = -10 * mdis.sum(3)
    if dskin is not None:
        mdis = mdis + dskin
    mdis = mdis - mdis.max(2, keepdims=True)
    e = np.exp(mdis)
    return (e / e.sum(2, keepdims=True)).astype(pts.dtype)


def dqs_blend_skinning(dq, skin, pts):
    """geom_utils.py:457-517: blend dual quaternions, normalise by the real norm, transform points."""
    B = dq.shape[-2]
    N = pts.shape[-2]
    pts = pts.reshape(-1, N, 3)
    dq = dq.reshape(-1, B, 8)
    b = np.einsum("snb,sbk->snk", skin, dq)  # (skin[...,None] * dq[:,None]).sum(2), geom_utils.py:470
    c = dq_normalize(b.astype(pts.dtype))
    a0, d0 = c[..., 0], c[..., 1:4]
    ae, de = c[..., 4], c[..., 5:8]
    trans = 2 * (a0[..., None] * de - ae[..., None] * d0 + np.cross(d0, de))
    rotated = pts + 2 * np.cross(d0, np.cross(d0, pts) + a0[..., None] * pts)
    return rotated + trans


def neu_dbs(bones, rts_fw, skin, xyz_in, backward=True):
    """geom_utils.py:372-456 without nerf_dis: returns the warped points only."""
    B = bones.shape[-2]
    N = xyz_in.shape[-2]
    xyz_in = xyz_in.reshape(-1, N, 3)
    rts_fw = rts_fw.reshape(-1, B, 8)
    dq = dq_inverse(rts_fw) if backward else rts_fw
    return dqs_blend_skinning(dq, skin, xyz_in)


# ----------------------------------------------------------------------------
# nnutils/rendering.py
# ----------------------------------------------------------------------------


def composite(rgbs, sigmas, feat, z_vals, rays_d, beta, noise=None, oob=None, vis_pred=None, rgb_filter_scale=0.0):
    """rendering.py:183-237: SDF->density, alpha, exclusive transmittance product, weighted sums.

    rgbs (N,S,3), sigmas (N,S) raw MLP sigma, feat (N,S,F), z_vals (N,S), rays_d (N,3).
    Returns rgb, feat, depth, weights, visibility, sil.
    """
    dt = z_vals.dtype
    deltas = z_vals[:, 1:] - z_vals[:, :-1]
    deltas = np.concatenate([deltas, np.full_like(deltas[:, :1], 1e10)], -1)
    deltas = deltas * np.sqrt((rays_d * rays_d).sum(-1))[:, None]
    semantic = dt.type(rgb_filter_scale) * _sigmoid(dt.type(-10) * sigmas)      # :171 (before the noise is added)
    if noise is not None:
        sigmas = sigmas + noise
    ibeta = dt.type(1) / (np.abs(dt.type(beta)) + dt.type(1e-9))
    sdf = -sigmas
    dens = (dt.type(0.5) + dt.type(0.5) * np.sign(sdf) * np.expm1(-np.abs(sdf) * ibeta)) * ibeta
    alphas = 1 - np.exp(-deltas * dens)
    if oob is not None:
        alphas = np.where(oob, dt.type(0), alphas)
    if vis_pred is not None:
        alphas = np.where(vis_pred < 0.5, dt.type(0), alphas)
    shifted = np.concatenate([np.ones_like(alphas[:, :1]), 1 - alphas + dt.type(1e-10)], -1)
    alpha_prod = np.cumprod(shifted, -1, dtype=dt)[:, :-1]
    weights = alphas * alpha_prod
    if rgb_filter_scale > 0:                                                     # opts.rgb_filter, :225-230
        rgb = ((weights[:, :-1] * semantic[:, :-1])[..., None] * rgbs[:, :-1]).sum(-2)
    else:
        rgb = (weights[..., None] * rgbs).sum(-2)
    ft = (weights[..., None] * feat).sum(-2)
    depth = (weights * z_vals).sum(-1)
    sil = weights[:, :-1].sum(-1)
    return rgb, ft, depth, weights, alpha_prod, sil


def sample_pdf(bins, weights, n_importance, u=None, eps=1e-5):
    """rendering.py:582-623.  u=None -> deterministic linspace (det=True); else injected uniforms."""
    dt = bins.dtype
    n_rays, n_s = weights.shape
    weights = weights + dt.type(eps)
    pdf = weights / weights.sum(-1, keepdims=True)
    cdf = np.cumsum(pdf, -1, dtype=dt)
    cdf = np.concatenate([np.zeros_like(cdf[:, :1]), cdf], -1)
    if u is None:
        u = np.broadcast_to(np.linspace(0, 1, n_importance, dtype=dt), (n_rays, n_importance))
    u = np.ascontiguousarray(u, dtype=dt)
    inds = np.stack([np.searchsorted(cdf[i], u[i], side="right") for i in range(n_rays)], 0)
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, n_s)
    cdf_b = np.take_along_axis(cdf, below, 1)
    cdf_a = np.take_along_axis(cdf, above, 1)
    bin_b = np.take_along_axis(bins, below, 1)
    bin_a = np.take_along_axis(bins, above, 1)
    denom = cdf_a - cdf_b
    denom = np.where(denom < eps, dt.type(1), denom)
    return bin_b + (u - cdf_b) / denom * (bin_a - bin_b)


def sample_z(near, far, n_samples, use_disp=False, perturb=0, perturb_rand=None):
    """rendering.py:68-83."""
    dt = near.dtype
    t = np.linspace(0, 1, n_samples, dtype=dt)
    if not use_disp:
        z = near * (1 - t) + far * t
    else:
        z = 1 / (1 / near * (1 - t) + 1 / far * t)
    z = np.broadcast_to(z, (near.shape[0], n_samples)).astype(dt)
    if perturb > 0:
        mid = dt.type(0.5) * (z[:, :-1] + z[:, 1:])
        upper = np.concatenate([mid, z[:, -1:]], -1)
        lower = np.concatenate([z[:, :1], mid], -1)
        z = lower + (upper - lower) * (dt.type(perturb) * perturb_rand.astype(dt))
    return z


class Scene:
    """Plain container: model parameters (state-dict-named arrays) + flags, the oracle's `models` dict."""

    def __init__(self, coarse, bones_rst=None, skin_aux=None, nerf_skin=None, rest_pose_code=None,
                 nerf_feat=None, nerf_vis=None, alpha_xyz=10, alpha_dir=4, n_freq_xyz=10, n_freq_dir=4, nerf_dis=None,
                 nerf_unc=None):
        self.coarse = coarse
        self.bones_rst = bones_rst
        self.skin_aux = skin_aux
        self.nerf_skin = nerf_skin
        self.rest_pose_code = rest_pose_code
        self.nerf_feat = nerf_feat
        self.nerf_vis = nerf_vis
        self.nerf_dis = nerf_dis
        self.nerf_unc = nerf_unc
        self.alpha_xyz, self.alpha_dir = alpha_xyz, alpha_dir
        self.n_freq_xyz, self.n_freq_dir = n_freq_xyz, n_freq_dir


def _mlp_dims(p):
    """Recover (D, W, in_xyz, in_dir, out) from state-dict shapes (nerf.py:109-135)."""
    D = sum(1 for k in p if k.startswith("xyz_encoding_") and k.endswith(".0.weight"))
    W, in_xyz = p["xyz_encoding_1.0.weight"].shape
    in_dir = p["dir_encoding.0.weight"].shape[1] - W
    out = p["rgb.0.weight"].shape[0]
    return D, W, in_xyz, in_dir, out


def gauss_mlp_skinning(scene, xyz, bones, pose_code, round_fn=None):
    """geom_utils.py:202-229: dskin = nerf_skin([PE(xyz), pose_code]); skin = softmax(gauss + dskin)."""
    n_rays = xyz.shape[0]
    dskin = None
    if scene.nerf_skin is not None:
        if pose_code.ndim == 2 and pose_code.shape[0] != n_rays:
            pose_code = np.broadcast_to(pose_code[None], (n_rays,) + pose_code.shape)
        D, W, in_xyz, in_dir, _ = _mlp_dims(scene.nerf_skin)
        fn = lambda x, sigma_only=False: nerf_forward(scene.nerf_skin, x, D=D, W=W, in_channels_xyz=in_xyz,
                                                     in_channels_dir=in_dir, raw_feat=True, round_fn=round_fn)
        emb = embedding(xyz, scene.n_freq_xyz, scene.alpha_xyz)
        dskin = evaluate_mlp(fn, emb, code=pose_code, chunk=8 * 1024)
    return skinning(bones, xyz, dskin, scene.skin_aux)


def residual_deformation(scene, xyz, code, round_fn=None):
    """calculate_residual_deformation (geom_utils.py:350-355): nerf_dis([PE(xyz), code]) -> (N,S,3)."""
    n_rays = xyz.shape[0]
    if code.ndim == 2 and code.shape[0] != n_rays:
        code = np.broadcast_to(code[None], (n_rays,) + code.shape)
    D, W, in_xyz, in_dir, _ = _mlp_dims(scene.nerf_dis)
    fn = lambda x, sigma_only=False: nerf_forward(scene.nerf_dis, x, D=D, W=W, in_channels_xyz=in_xyz,
                                                 in_channels_dir=in_dir, raw_feat=True, round_fn=round_fn)
    return evaluate_mlp(fn, embedding(xyz, scene.n_freq_xyz, scene.alpha_xyz), code=code, chunk=8 * 1024)


def inference_deform(scene, xyz_sampled, rays, z_vals, dir_embedded, fine_iter=True, render_vis=False,
                     obj_bound=None, symm_mask=None, noise=None, round_fn=None, rgb_filter_scale=0.0):
    """rendering.py:239-579, bones/neudbs branch + plain-NeRF branch, without the loss heads."""
    rays_d = rays["rays_d"]
    n_rays, n_samples = z_vals.shape
    result = {}
    xyz_frame = xyz_sampled.copy()
    xyz = xyz_sampled
    frame_cyc_dis = None
    if scene.bones_rst is not None:
        bones_rst = scene.bones_rst
        bone_rts_fw = rays["bone_rts"]
        time_embedded = rays["time_embedded"][:, None]
        bones_dfm = bone_transform(bones_rst, bone_rts_fw)                      # rendering.py:303
        skin_bw = gauss_mlp_skinning(scene, xyz, bones_dfm, time_embedded, round_fn)   # :304
        xyz_in = xyz
        xyz = neu_dbs(bones_rst, bone_rts_fw, skin_bw, xyz, backward=True)      # :319
        has_dis = getattr(scene, "nerf_dis", None) is not None
        if has_dis:                                                             # geom_utils.py:416-418, rendering.py:321-322
            xyz_dis = residual_deformation(scene, xyz_in, time_embedded, round_fn)
            xyz = xyz - xyz_dis
            result["dis_reg"] = np.sqrt((xyz_dis * xyz_dis).sum(-1))
        if fine_iter:
            skin_fw = gauss_mlp_skinning(scene, xyz, bones_rst, scene.rest_pose_code, round_fn)  # :330
            xyz_tf = xyz
            if has_dis:                                                         # geom_utils.py:420-422, rendering.py:342-343
                dis_f = residual_deformation(scene, xyz, scene.rest_pose_code, round_fn)
                xyz_tf = xyz + dis_f
                result["dis_reg_forward"] = np.sqrt((dis_f * dis_f).sum(-1))
            xyz_cyc = neu_dbs(bones_rst, bone_rts_fw, skin_fw, xyz_tf, backward=False)    # :338
            d = xyz_frame - xyz_cyc
            frame_cyc_dis = np.sqrt((d * d).sum(-1))                            # :341
    env_code = rays.get("env_code")
    app_code = rays.get("appearance_code")
    oob = None
    vis_pred = None
    if render_vis:                                                              # :375-379
        D, W, in_xyz, in_dir, _ = _mlp_dims(scene.nerf_vis)
        fnv = lambda x, sigma_only=False: nerf_forward(scene.nerf_vis, x, D=D, W=W, in_channels_xyz=in_xyz,
                                                      in_channels_dir=in_dir, raw_feat=True, round_fn=round_fn)
        vis_pred = _sigmoid(evaluate_mlp(fnv, embedding(xyz, scene.n_freq_xyz, scene.alpha_xyz))[..., 0])
    if symm_mask is not None:                                                   # :385-391
        xyz_x = np.where(symm_mask, -xyz[..., :1], xyz[..., :1])
        xyz_input = np.concatenate([xyz_x, xyz[..., 1:3]], -1)
    else:
        xyz_input = xyz
    if render_vis and obj_bound is not None:                                    # :210-213
        cb = np.asarray(obj_bound, dtype=xyz.dtype).reshape(1, 1, 3)
        oob = (np.abs(xyz_input) > cb).sum(-1) > 0
    # inference(): rendering.py:124-237
    D, W, in_xyz, in_dir, _ = _mlp_dims(scene.coarse)
    fnc = lambda x, sigma_only=False: nerf_forward(scene.coarse, x, D=D, W=W, in_channels_xyz=in_xyz,
                                                  in_channels_dir=in_dir, sigma_only=sigma_only, round_fn=round_fn)
    emb_fn = lambda x: embedding(x, scene.n_freq_xyz, scene.alpha_xyz)
    dir_e = np.broadcast_to(dir_embedded[:, None], (n_rays, n_samples, dir_embedded.shape[-1]))
    out = evaluate_mlp(fnc, xyz_input, embed_fn=emb_fn, dir_embedded=dir_e, code=env_code,
                       appearance_code=app_code, chunk=4096)
    rgbs, sigmas = out[..., :3], out[..., 3]
    if scene.nerf_feat is not None:
        Df, Wf, in_xyz_f, in_dir_f, _ = _mlp_dims(scene.nerf_feat)
        fnf = lambda x, sigma_only=False: nerf_forward(scene.nerf_feat, x, D=Df, W=Wf, in_channels_xyz=in_xyz_f,
                                                      in_channels_dir=in_dir_f, raw_feat=True, round_fn=round_fn)
        feat = evaluate_mlp(fnf, xyz_input, embed_fn=emb_fn, chunk=4096)
    else:
        feat = np.zeros_like(rgbs)
    rgb, feat_rnd, depth, weights, vis, sil = composite(
        rgbs, sigmas, feat, z_vals, rays_d, scene.coarse["beta"][0], noise=noise, oob=oob, vis_pred=vis_pred,
        rgb_filter_scale=rgb_filter_scale)
    result["img_coarse"] = rgb
    result["depth_rnd"] = depth
    result["sil_coarse"] = weights[:, :-1].sum(1)
    result["feat_rnd"] = feat_rnd
    result["weights_coarse"] = weights
    result["vis_coarse"] = vis
    if render_vis:
        result["vis_pred"] = (vis_pred * weights).sum(-1)
    if fine_iter:
        result["xyz_camera_vis"] = xyz_frame
        if scene.bones_rst is not None:
            result["xyz_canonical_vis"] = xyz
            result["frame_cyc_dis"] = (frame_cyc_dis * weights).sum(-1)
        if getattr(scene, "nerf_unc", None) is not None:                        # rendering.py:501-516, nerf.py:502-511
            xyt = np.concatenate([rays["xysn"], rays["ts"]], -1)
            x = np.concatenate([embedding(xyt, scene.n_freq_xyz, scene.alpha_xyz), rays["vid_code"]], -1)
            Du, Wu, in_xyz_u, in_dir_u, _ = _mlp_dims(scene.nerf_unc)
            result["unc_pred"] = nerf_forward(scene.nerf_unc, x, D=Du, W=Wu, in_channels_xyz=in_xyz_u,
                                              in_channels_dir=in_dir_u, raw_feat=True, round_fn=round_fn)
    return result, weights


def render_rays(scene, rays, N_samples=64, use_disp=False, perturb=0, use_fine=False, render_vis=False,
                obj_bound=None, perturb_rand=None, pdf_u=None, symm_mask=None, symm_mask_pre=None,
                noise=None, noise_pre=None, round_fn=None, rgb_filter_scale=0.0):
    """rendering.py:19-122.  rgb_filter_scale > 0: opts.rgb_filter with scale_rgb (:229-230).  Random draws are injected: perturb_rand (:82), pdf_u (:607),
    symm_mask / symm_mask_pre (:389, final / pre-pass), noise / noise_pre (:193, already scaled by noise_std)."""
    if use_fine:
        N_samples = N_samples // 2
    rays_o, rays_d, near, far = rays["rays_o"], rays["rays_d"], rays["near"], rays["far"]
    dt = rays_d.dtype
    d_norm = rays_d / np.sqrt((rays_d * rays_d).sum(-1))[:, None]
    dir_embedded = embedding(d_norm, scene.n_freq_dir, scene.alpha_dir)
    z_vals = sample_z(near, far, N_samples, use_disp, perturb, perturb_rand)
    xyz = rays_o[:, None] + rays_d[:, None] * z_vals[:, :, None]
    if use_fine:
        _, w = inference_deform(scene, xyz, rays, z_vals, dir_embedded, fine_iter=False,
                                symm_mask=symm_mask_pre, noise=noise_pre, round_fn=round_fn, rgb_filter_scale=rgb_filter_scale)
        mid = dt.type(0.5) * (z_vals[:, :-1] + z_vals[:, 1:])
        z_new = sample_pdf(mid, w[:, 1:-1], N_samples, u=(None if perturb == 0 else pdf_u))
        z_vals = np.sort(np.concatenate([z_vals, z_new], -1), -1)
        xyz = rays_o[:, None] + rays_d[:, None] * z_vals[:, :, None]
    result, _ = inference_deform(scene, xyz, rays, z_vals, dir_embedded, fine_iter=True, render_vis=render_vis,
                                 obj_bound=obj_bound, symm_mask=symm_mask, noise=noise, round_fn=round_fn,
                                 rgb_filter_scale=rgb_filter_scale)
    result["z_vals"] = z_vals
    return result


# ---------------------------------------------------------------------------------------------------------------------
# S3IM (opts.s3im_loss; reference nnutils/loss_utils.py:575-702, called from nnutils/rendering.py:528-532)
def s3im_window(kernel_size=4, sigma=1.5):
    """loss_utils.py:575-583 gaussian / create_window: g[x] = exp(-(x - k//2)^2 / (2 sigma^2)) normalised, outer product."""
    g = np.asarray([np.exp(-(x - kernel_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(kernel_size)], np.float64)
    g = g / g.sum()
    return np.outer(g, g)


def s3im_index(perms, patch=32 * 32):
    """loss_utils.py:677-684: the identity followed by the random permutations the reference draws (torch.randperm)."""
    return np.concatenate([np.arange(patch)] + [np.asarray(p, np.int64) for p in perms])


def s3im_loss(src, tar, mask, perms, kernel_size=4, stride=4, patch_h=32, patch_w=32):
    """S3IM.forward (loss_utils.py:662-702) with SSIM(window 4, stride 4) = _ssim (:585-608) -> scalar (float64).
    src, tar (N,3), mask (N,1): both are multiplied by the mask (:665-666), cut or tiled to patch_h * patch_w rows (:670-677),
    gathered through [identity, perms...] (:678-686), laid out as a (3, patch_h, patch_w * repeats) image (:691-692: a plain
    reshape of the (3, P * R) matrix), and compared by the mean SSIM of conv2d(padding (k-1)//2 = 1, stride 4) windows."""
    src = np.asarray(src, np.float64) * np.asarray(mask, np.float64)
    tar = np.asarray(tar, np.float64) * np.asarray(mask, np.float64)
    P = patch_h * patch_w
    n = src.shape[0]
    rows = np.arange(P) % n                      # n >= P: the first P rows; n < P: repeat(...)[:P]  (:670-677)
    idx = rows[s3im_index(perms, P)]
    R = idx.shape[0] // P
    a = src[idx].T.reshape(3, patch_h, patch_w * R)
    b = tar[idx].T.reshape(3, patch_h, patch_w * R)
    w = s3im_window(kernel_size)
    pad = (kernel_size - 1) // 2

    def conv(x):
        xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad)))
        oh = (xp.shape[1] - kernel_size) // stride + 1
        ow = (xp.shape[2] - kernel_size) // stride + 1
        out = np.zeros((3, oh, ow))
        for i in range(kernel_size):
            for j in range(kernel_size):
                out += w[i, j] * xp[:, i:i + stride * oh:stride, j:j + stride * ow:stride]
        return out
    mu1, mu2 = conv(a), conv(b)
    s1 = conv(a * a) - mu1 ** 2
    s2 = conv(b * b) - mu2 ** 2
    s12 = conv(a * b) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))
    return 1.0 - ssim.mean()
