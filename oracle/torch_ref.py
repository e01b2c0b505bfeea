"""Differentiable CPU restatement of the rendering path in PyTorch (autograd) -- TEST INFRASTRUCTURE ONLY.

Same maths and the same reference citations as oracle/moda_oracle.py (the numpy oracle), written with torch ops
so that autograd supplies reference GRADIENTS for the backward kernels.  Pinned two ways in the CPU suite
(tests/test_torch_ref.py): its forward outputs against the numpy oracle / the reference's golden outputs, and its
gradients against golden gradients produced by the reference's own autograd (tests/golden/g9_grad_*.npz).
Only tests/ may import it; the product package never does.
"""
import math

import torch
import torch.nn.functional as F


def _q_raw_mul(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz,
                        aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_to_matrix(q):
    r, i, j, k = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def dq_inverse(dq):
    """dual_quat.py:87-94"""
    conj = dq * dq.new_tensor([1, -1, -1, -1, 1, -1, -1, -1])
    return conj / (dq[..., :4] ** 2).sum(-1, keepdim=True)


def embedding(x, n_freqs, alpha=None):
    """nerf.py:35-75"""
    alpha = n_freqs if alpha is None else float(alpha)
    out = [x]
    for k in range(n_freqs):
        w = min(max(alpha - k, 0.0), 1.0)
        win = 0.5 * (1 + math.cos(math.pi * w + math.pi))
        out.append(win * torch.sin((2.0 ** k) * x))
        out.append(win * torch.cos((2.0 ** k) * x))
    return torch.cat(out, -1)


class _Bf16OperandLinear(torch.autograd.Function):
    """y = bf16(x) bf16(W)^T + b with fp32 products and sums, and the same operand rounding in the backward GEMMs
    (dx = bf16(dy) bf16(W), dW = bf16(dy)^T bf16(x)): what the training route's throughput mode computes
    (MODA_GEMM_BF16, moda_amd.set_train_precision('bf16'))."""

    @staticmethod
    def forward(ctx, x, w, b):
        r = lambda t: t.to(torch.bfloat16).to(torch.float32)
        ctx.save_for_backward(x, w)
        return r(x) @ r(w).T + b

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        r = lambda t: t.to(torch.bfloat16).to(torch.float32)
        gr = r(g)
        g2, x2 = gr.reshape(-1, gr.shape[-1]), r(x).reshape(-1, x.shape[-1])
        return gr @ r(w), g2.T @ x2, g.reshape(-1, g.shape[-1]).sum(0)


LINEAR_BF16_OPERANDS = False      # tests switch this on to obtain the rounding oracle of the bf16 training mode


def _lin(t, w, b):
    if LINEAR_BF16_OPERANDS:
        return _Bf16OperandLinear.apply(t, w, b)
    return F.linear(t, w, b)                 # what nn.Linear (nerf.py:109-140) calls: one addmm, no separate bias pass


# Conditioning probe (tests): when RELU_MARGINS is a list, every ReLU pre-activation tensor z appends (rows, per-row min of
# |z| / max|z|) -- in float64 this says which samples hold a ReLU whose state two correct fp32 evaluations may disagree on.
RELU_MARGINS = None


def _note_margin(z):
    if RELU_MARGINS is not None:
        a = z.detach().reshape(-1, z.shape[-1]).abs()
        RELU_MARGINS.append((a / a.max()).min(-1).values)
    return z


def nerf_forward(p, x, D, W, in_xyz, in_dir, raw_feat=False, sigma_only=False):
    """nerf.py:147-198; p maps state-dict names to tensors."""
    input_xyz = x[..., :in_xyz]
    input_dir = x[..., in_xyz:in_xyz + in_dir]
    h = input_xyz
    lin = lambda t, n: _lin(t, p[n + ".weight"], p[n + ".bias"])
    for i in range(D):
        if i == 4:
            h = torch.cat([input_xyz, h], -1)
        h = torch.relu_(_note_margin(lin(h, f"xyz_encoding_{i+1}.0")))   # nn.ReLU(True): in place, as the reference (nerf.py:86)
    sigma = lin(h, "sigma")
    if sigma_only:
        return sigma
    final = lin(h, "xyz_encoding_final")
    d = torch.relu_(_note_margin(lin(torch.cat([final, input_dir], -1), "dir_encoding.0")))
    rgb = lin(d, "rgb.0")
    return rgb if raw_feat else torch.cat([torch.sigmoid(rgb), sigma], -1)


def _dims(p):
    D = sum(1 for k in p if k.startswith("xyz_encoding_") and k.endswith(".0.weight"))
    W, in_xyz = p["xyz_encoding_1.0.weight"].shape
    return D, W, in_xyz, p["dir_encoding.0.weight"].shape[1] - W


def bone_transform(bones, rts):
    """geom_utils.py:59-111 (neudbs)"""
    B = bones.shape[-2]
    rts = rts.reshape(-1, B, 8)
    dq_r, dq_d = rts[..., :4], rts[..., 4:]
    R = quaternion_to_matrix(dq_r)
    inv = dq_r * dq_r.new_tensor([1, -1, -1, -1])
    t = 2 * _q_raw_mul(dq_d, inv)[..., 1:]
    center = (R @ bones[None, :, :3, None])[..., 0] + t
    orient = _q_raw_mul(dq_r, bones[None, :, 3:7].expand(rts.shape[0], B, 4))
    orient = torch.where(orient[..., :1] < 0, -orient, orient)
    scale = bones[None, :, 7:10].expand(rts.shape[0], B, 3)
    return torch.cat([center, orient, scale], -1)


def skinning(bones, pts, dskin, skin_aux):
    """geom_utils.py:237-302"""
    bs, N, _ = pts.shape
    B = bones.shape[-2]
    if bones.dim() == 2:
        bones = bones[None].expand(bs, B, 10)
    q = bones[..., 3:7]
    q = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    Rt = quaternion_to_matrix(q).transpose(-1, -2)
    scale = bones[..., 7:10].exp()
    mdis = bones[:, None, :, :3] - pts[:, :, None, :]
    mdis = (Rt[:, None] * mdis[..., None, :]).sum(-1)
    mdis = scale[:, None] * mdis ** 2
    mdis = mdis * 100 * skin_aux[0].exp()
    logits = -10 * mdis.sum(-1)
    if dskin is not None:
        logits = logits + dskin
    return logits.softmax(-1)


def dqs(dq, skin, pts):
    """geom_utils.py:457-517"""
    b = torch.einsum("snb,sbk->snk", skin, dq)
    c = b / b[..., :4].norm(dim=-1, keepdim=True)
    a0, d0, ae, de = c[..., 0:1], c[..., 1:4], c[..., 4:5], c[..., 5:8]
    trans = 2 * (a0 * de - ae * d0 + torch.cross(d0, de, dim=-1))
    return pts + 2 * torch.cross(d0, torch.cross(d0, pts, dim=-1) + a0 * pts, dim=-1) + trans


def composite(rgbs, sigmas, feat, z_vals, rays_d, beta, noise=None, rgb_filter_scale=0.0):
    """rendering.py:183-237; rgb_filter_scale > 0 is opts.rgb_filter with scale_rgb (:171, 225-230)"""
    deltas = z_vals[:, 1:] - z_vals[:, :-1]
    deltas = torch.cat([deltas, torch.full_like(deltas[:, :1], 1e10)], -1) * rays_d.norm(dim=-1, keepdim=True)
    semantic = rgb_filter_scale * torch.sigmoid(-10 * sigmas)
    if noise is not None:
        sigmas = sigmas + noise
    ibeta = 1 / (beta.abs() + 1e-9)
    sdf = -sigmas
    dens = (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() * ibeta)) * ibeta
    alphas = 1 - torch.exp(-deltas * dens)
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    T = torch.cumprod(shifted, -1)[:, :-1]
    w = alphas * T
    if rgb_filter_scale > 0:
        rgb = ((w[:, :-1] * semantic[:, :-1])[..., None] * rgbs[:, :-1]).sum(-2)
    else:
        rgb = (w[..., None] * rgbs).sum(-2)
    return (rgb, (w[..., None] * feat).sum(-2), (w * z_vals).sum(-1), w, T.detach(), w[:, :-1].sum(-1))


def render_rays(m, rays, N_samples, alpha=10.0, noise=None, rgb_filter_scale=0.0):
    """rendering.py:19-122, 239-579 for use_fine=False, perturb=0, fine_iter=True, no loss heads.
    m: dict with 'coarse' (state-dict tensors), optional 'bones_rst', 'skin_aux', 'nerf_skin', 'rest_pose_code'."""
    rays_o, rays_d, near, far = rays["rays_o"], rays["rays_d"], rays["near"], rays["far"]
    N = rays_d.shape[0]
    d_norm = rays_d / rays_d.norm(dim=-1, keepdim=True)
    dir_emb = embedding(d_norm, 4, alpha)
    t = torch.linspace(0, 1, N_samples, dtype=rays_d.dtype)
    z = (near * (1 - t) + far * t).expand(N, N_samples)
    xyz = rays_o[:, None] + rays_d[:, None] * z[..., None]
    xyz_frame = xyz
    res = {}
    cyc = None
    if "bones_rst" in m:
        bones, rts, aux = m["bones_rst"], rays["bone_rts"], m["skin_aux"]
        B = bones.shape[0]
        bones_dfm = bone_transform(bones, rts)

        def dskin_of(pts, code):
            if "nerf_skin" not in m:
                return None
            D, W, in_xyz, in_dir = _dims(m["nerf_skin"])
            x = torch.cat([embedding(pts, 10, alpha), code.expand(N, N_samples, code.shape[-1])], -1)
            return nerf_forward(m["nerf_skin"], x, D, W, in_xyz, in_dir, raw_feat=True)

        def dis_of(pts, code):                              # calculate_residual_deformation, geom_utils.py:350-355
            D, W, in_xyz, in_dir = _dims(m["nerf_dis"])
            x = torch.cat([embedding(pts, 10, alpha), code.expand(N, N_samples, code.shape[-1])], -1)
            return nerf_forward(m["nerf_dis"], x, D, W, in_xyz, in_dir, raw_feat=True)

        rest = m.get("rest_pose_code", torch.zeros(1, 128))[None]
        skin_bw = skinning(bones_dfm, xyz, dskin_of(xyz, rays["time_embedded"][:, None]), aux)
        xyz = dqs(dq_inverse(rts.reshape(N, B, 8)), skin_bw, xyz)
        if "nerf_dis" in m:                                 # geom_utils.py:416-418, rendering.py:321-322
            xyz_dis = dis_of(xyz_frame, rays["time_embedded"][:, None])
            xyz = xyz - xyz_dis
            res["dis_reg"] = xyz_dis.norm(dim=2)
        skin_fw = skinning(bones, xyz, dskin_of(xyz, rest), aux)
        xyz_tf = xyz
        if "nerf_dis" in m:                                 # geom_utils.py:420-422, rendering.py:342-343
            dis_f = dis_of(xyz, rest)
            xyz_tf = xyz + dis_f
            res["dis_reg_forward"] = dis_f.norm(dim=2)
        res["_skin_fw"], res["_xyz_tf"] = skin_fw, xyz_tf
        xyz_cyc = dqs(rts.reshape(N, B, 8), skin_fw, xyz_tf)
        cyc = (xyz_frame - xyz_cyc).norm(dim=-1)
    D, W, in_xyz, in_dir = _dims(m["coarse"])
    side = [dir_emb[:, None].expand(N, N_samples, 27), rays["env_code"][:, None].expand(N, N_samples, 64)]
    if "appearance_code" in rays:                          # rendering.py:369-372, geom_utils.py:45-50
        side.append(rays["appearance_code"][:, None].expand(N, N_samples, rays["appearance_code"].shape[-1]))
    x = torch.cat([embedding(xyz, 10, alpha)] + side, -1)
    out = nerf_forward(m["coarse"], x, D, W, in_xyz, in_dir)
    feat = torch.zeros_like(out[..., :3])
    if "nerf_feat" in m:                                   # rendering.py:174-178
        Df, Wf, in_f, dir_f = _dims(m["nerf_feat"])
        feat = nerf_forward(m["nerf_feat"], embedding(xyz, 10, alpha), Df, Wf, in_f, dir_f, raw_feat=True)
    rgb, feat_rnd, depth, w, vis, sil = composite(out[..., :3], out[..., 3], feat, z, rays_d, m["coarse"]["beta"], noise,
                                                  rgb_filter_scale)
    res.update(img_coarse=rgb, depth_rnd=depth, sil_coarse=sil, xyz_camera_vis=xyz_frame, weights=w, visibility=vis,
               feat_rnd=feat_rnd)
    if cyc is not None:
        res["xyz_canonical_vis"] = xyz
        res["frame_cyc_dis"] = (cyc * w.detach()).sum(-1)
    if "nerf_unc" in m:                                    # rendering.py:501-516, nerf.py:502-511
        Du, Wu, in_u, dir_u = _dims(m["nerf_unc"])
        xu = torch.cat([embedding(torch.cat([rays["xysn"], rays["ts"]], -1), 10, alpha), rays["vid_code"]], -1)
        res["unc_pred"] = nerf_forward(m["nerf_unc"], xu, Du, Wu, in_u, dir_u, raw_feat=True)
    return res


# ---- loss heads behind compositing (rendering.py:410-437, 475-477, 573-578; nnutils/loss_utils.py) --------------
def normalize(x):
    """F.normalize(x, 2, -1)"""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def project(pts, rtk_vec):
    """obj_to_cam + pinhole_cam with K = mat2K(Kmatinv(Kinv)) (geom_utils.py:567-581, 612-673; rendering.py:443-449)."""
    N = pts.shape[0]
    R = rtk_vec[:, 0:9].reshape(N, 1, 3, 3)
    Tm = rtk_vec[:, 9:12].reshape(N, 1, 3)
    Ki = rtk_vec[:, 12:21].reshape(N, 3, 3)
    fx, fy = 1.0 / Ki[:, 0, 0], 1.0 / Ki[:, 1, 1]
    px, py = -Ki[:, 0, 2] * fx, -Ki[:, 1, 2] * fy
    cam = (R * pts[:, :, None, :]).sum(-1) + Tm
    z = cam[..., 2]
    u = (fx[:, None] * cam[..., 0] + px[:, None] * z) / (1e-6 + z)
    v = (fy[:, None] * cam[..., 1] + py[:, None] * z) / (1e-6 + z)
    return torch.stack([u, v, z], -1)


def forward_warp(m, pts, rts, alpha=10.0):
    """gauss_mlp_skinning with the rest-pose code + neu_dbs(backward=False) (loss_utils.py:250-254)."""
    N, n = pts.shape[:2]
    bones, aux = m["bones_rst"], m["skin_aux"]
    B = bones.shape[0]
    ds = None
    if "nerf_skin" in m:
        D, W, in_xyz, in_dir = _dims(m["nerf_skin"])
        code = m["rest_pose_code"][None].expand(N, n, 128)
        ds = nerf_forward(m["nerf_skin"], torch.cat([embedding(pts, 10, alpha), code], -1), D, W, in_xyz, in_dir, raw_feat=True)
    return dqs(rts.reshape(N, B, 8), skinning(bones, pts, ds, aux), pts)


def query_grid(bound, grid_size=20):
    """loss_utils.py:290-294: (x_i, y_j, z_k), C-order over (i, j, k)."""
    # the reference's expression on the caller's scalars (a float32 `bound` makes NumPy >= 2 compute the nodes in float32)
    import numpy as np
    ax = [torch.from_numpy(np.linspace(-bound[c], bound[c], grid_size).astype(np.float32)) for c in range(3)]
    g = torch.stack(torch.meshgrid(ax[0], ax[1], ax[2], indexing="ij"), -1)
    return g.reshape(-1, 3)


def feat_match(m, feats, bound, use_ot, noise=None, alpha=10.0, use_corr=False):
    """loss_utils.py:273-405: pixel features (n,16) -> expected canonical location (n,3) [, corr_err (n) with use_corr]."""
    fn = normalize(feats)
    dt = feats.dtype                                             # float64 when the caller evaluates the truth (tests: *_f64 fixtures)
    query = query_grid(bound).to(dt)
    if noise is not None:                                        # :304-306 (training only)
        query = query + noise.reshape(query.shape).to(dt) * torch.as_tensor(bound, dtype=torch.float32).to(dt) * 0.05
    D, W, in_xyz, in_dir = _dims(m["nerf_feat"])
    vol = normalize(nerf_forward(m["nerf_feat"], embedding(query, 10, alpha), D, W, in_xyz, in_dir, raw_feat=True))
    cost = fn @ vol.T
    if use_ot:                                                   # :338-374
        K = torch.exp(-(1.0 - cost) / 0.03)
        a = torch.full((K.shape[0], 1), 1.0 / K.shape[0], dtype=dt)
        p1, p2 = a.clone(), torch.full((K.shape[1], 1), 1.0 / K.shape[1], dtype=dt)
        for _ in range(20):
            b = p2 / (K.T @ a + 1e-8)
            a = p1 / (K @ b + 1e-8)
        Tm = a * K * b.T
        prob = Tm / Tm.sum(1, keepdim=True)
    else:                                                        # :331-332, :376
        prob = (cost * (m["nerf_feat"]["beta"].abs() + 1e-9)).softmax(-1)
    if use_corr:                                                 # :386-391
        tt = prob @ prob.T
        return prob @ query, (tt - torch.eye(tt.shape[0], dtype=tt.dtype)).norm(2, -1)
    return prob @ query


def visibility_loss(m, xyz_pos, w_pos, bound, neg_rand, alpha=10.0):
    """loss_utils.py:125-149"""
    import torch.nn.functional as F
    D, W, in_xyz, in_dir = _dims(m["nerf_vis"])
    n = w_pos.numel()
    bnd = torch.as_tensor(bound, dtype=torch.float32)[None, None].to(xyz_pos.dtype)
    xyz_neg = neg_rand.reshape(1, n, 3).to(xyz_pos.dtype) * 2 * bnd - bnd
    f = lambda x: nerf_forward(m["nerf_vis"], embedding(x, 10, alpha), D, W, in_xyz, in_dir, raw_feat=True)[..., 0]
    neg = -F.logsigmoid(-f(xyz_neg)).sum() * 0.1 / n
    pos = -(F.logsigmoid(f(xyz_pos.detach())) * w_pos.detach()).sum() / n
    return pos + neg


def feature_heads(m, rays, res, bound, use_ot, img_size, feat_noise=None, vis_neg_rand=None, training=True, alpha=10.0,
                  use_corr=False):
    """rendering.py:417-437, 475-477, 573-578 on the outputs of render_rays() above."""
    w, xyz = res["weights"], res["xyz_canonical_vis"]
    N = w.shape[0]
    out = {}
    pw = w / (1e-9 + w.sum(1, keepdim=True))
    out["pts_exp"] = (xyz * pw[..., None]).sum(1)                # compute_pts_exp, loss_utils.py:165-175
    fm = feat_match(m, rays["feats_at_samp"], bound, use_ot, feat_noise if training else None, alpha, use_corr)
    if use_corr:
        out["pts_pred"], ce = fm
        out["corr_err"] = ce[:, None]                             # rendering.py:434-435
    else:
        out["pts_pred"] = fm
    out["feat_err"] = (out["pts_pred"] - out["pts_exp"]).norm(dim=-1, keepdim=True)
    pts = out["pts_pred"].reshape(N, 1, 3)
    if "bones_rst" in m:
        pts = forward_warp(m, pts, rays["bone_rts"], alpha)
    xy = project(pts, rays["rtk_vec"])[..., :2]
    out["proj_err"] = (rays["xys"].reshape(N, 1, 2) - xy).norm(dim=-1) / img_size * 2
    if training and "nerf_vis" in m:
        out["vis_loss"] = visibility_loss(m, xyz, res["visibility"], bound, vis_neg_rand, alpha)
    frnd = (normalize(res["feat_rnd"]) - rays["feats_at_samp"]).pow(2).mean(-1)
    out["frnd_loss_samp"] = frnd * rays["sil_at_samp"][..., 0]
    return out


def eikonal_loss(p, pts, bound, ppr, beta=0.1, alpha=10.0, eps=1e-3):
    """loss_utils.py:73-104 on the density head (p: nerf_coarse state dict, leaves may require grad): the analytic form
    differentiates the backward pass (nerf_gradient :15-47), the `ppr` form is compute_gradients_sdf :48-71."""
    D, W, in_xyz, in_dir = _dims(p)
    pts = pts.reshape(-1, 3).detach()
    inb = ((torch.as_tensor(bound, dtype=pts.dtype)[None] - pts.abs()) > 0).sum(-1) == 3
    pts = pts[inb]
    f = lambda x: nerf_forward(p, embedding(x, 10, alpha), D, W, in_xyz, in_dir, sigma_only=True)
    if ppr:
        ks = [pts.new_tensor(k) for k in ((1, -1, -1), (-1, -1, 1), (-1, 1, -1), (1, 1, 1))]
        g = sum(k * f(pts + k * eps) for k in ks) / (4.0 * eps)
    else:
        x = pts.clone().requires_grad_(True)
        y = f(x)
        g = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)[0]
    return ((g.norm(2, dim=-1) - 1) ** 2).mean()


def s3im_loss(src, tar, mask, perms, kernel_size=4, stride=4, patch_h=32, patch_w=32):
    """S3IM.forward + SSIM (reference loss_utils.py:585-608, 662-702), differentiable w.r.t. src; see oracle/moda_oracle.py
    s3im_loss for the line-by-line notes.  perms: (R-1, P) integer tensor of the permutations after the identity."""
    import torch.nn.functional as F
    src = src * mask
    tar = tar * mask
    P = patch_h * patch_w
    n = src.shape[0]
    rows = torch.arange(P) % n
    index = torch.cat([torch.arange(P)] + [p.long() for p in perms])
    idx = rows[index]
    R = idx.shape[0] // P
    a = src[idx].permute(1, 0).reshape(1, 3, patch_h, patch_w * R)
    b = tar[idx].permute(1, 0).reshape(1, 3, patch_h, patch_w * R)
    g = torch.tensor([math.exp(-(x - kernel_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(kernel_size)], dtype=src.dtype)
    g = g / g.sum()
    w = (g[:, None] @ g[None, :]).expand(3, 1, kernel_size, kernel_size).contiguous()
    conv = lambda x: F.conv2d(x, w, padding=(kernel_size - 1) // 2, groups=3, stride=stride)
    mu1, mu2 = conv(a), conv(b)
    s1 = conv(a * a) - mu1 ** 2
    s2 = conv(b * b) - mu2 ** 2
    s12 = conv(a * b) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))
    return 1 - ssim.mean()


def total_loss(rendered, w):
    """moda.py:540-705, default branches: the weighted means over boolean selections, summed (vis_loss with its own weight)."""
    sil = rendered["sil_at_samp"]
    t = {}
    t["img"] = w["img_wt"] * rendered["img_loss_samp"][sil[..., 0] > 0].mean()
    t["sil"] = (w["sil_wt"] * rendered["sil_loss_samp"])[rendered["vis_at_samp"] > 0].mean()
    t["frnd"] = (w["frnd_wt"] * rendered["frnd_loss_samp"])[sil[..., 0] > 0].mean()
    if "flo_loss_samp" in rendered:
        t["flo"] = rendered["flo_loss_samp"][rendered["sil_at_samp_flo"][..., 0]].mean() * 2 * w["flow_wt"]
    if "feat_err" in rendered:
        t["feat"] = (rendered["feat_err"] * w["feat_wt"])[sil > 0].mean()
    if "proj_err" in rendered:
        t["proj"] = (rendered["proj_err"] * w["proj_wt"])[sil > 0].mean()
    if "vis_loss" in rendered:
        t["vis"] = w["vis_wt"] * rendered["vis_loss"].mean()
    if "frame_cyc_dis" in rendered:
        t["cyc"] = rendered["frame_cyc_dis"].mean() * w["cyc_wt"]
    total = 0
    for v in t.values():
        total = total + v
    return total, t
