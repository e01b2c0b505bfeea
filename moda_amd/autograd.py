"""torch.autograd.Functions over the HIP library: the training route of the rendering path.

The reference trains through PyTorch autograd on eager ops.  Here every arithmetic-heavy node is one Function whose
forward and backward are HIP kernels (exact fp32, MFMA GEMMs for the layers); torch's own autograd only stitches
them together through data-movement ops (cat / expand / slicing, whose backward is a copy or a segment sum) and the
per-(ray, bone) preparation of the skinning data (a few dozen flops on N*B elements, `bone_prep` / `bone_transform`).
Gradient parity is pinned against the reference's own autograd (tests/golden/g9_grad_*.npz).
"""
import torch
from torch.autograd import Function

from . import _lib as L


def _f32(t):
    return L.dev(t)


def gemm(a, b, bias=None, act=0, mask_src=None, out=None, accumulate=False, split_k=1):
    """out (M,N) = act(a @ b + bias) for fp32 CUDA matrices with arbitrary strides (views are fine)."""
    M, K = a.shape
    K2, N = b.shape
    assert K == K2
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    L.call("moda_gemm_f32", L.ptr(a), a.stride(0), a.stride(1), L.ptr(b), b.stride(0), b.stride(1), L.ptr(out),
           out.stride(0), M, N, K, L.ptr(bias), act, L.ptr(mask_src), int(accumulate), int(split_k), L.stream())
    return out


def _as2d(x):
    x = x if x.dtype == torch.float32 else x.float()
    if x.dim() != 2:
        x = x.reshape(-1, x.shape[-1])
    if x.stride(1) != 1 and x.stride(0) != 1:
        x = x.contiguous()
    return x


class LinearFn(Function):
    """y = act(x W^T + b)  (nn.Linear + ReLU / sigmoid of nerf.py:111-135, 176-193)."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        x2 = _as2d(x)
        Wc = _f32(W)
        y = gemm(x2, Wc.t(), bias=_f32(b), act=act)
        ctx.act = act
        ctx.save_for_backward(x2, Wc, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, W, y = ctx.saved_tensors
        dy = _f32(dy)
        if ctx.act:
            dz = torch.empty_like(dy)
            L.call("moda_act_bwd", L.ptr(dy), L.ptr(y), dy.numel(), ctx.act, L.ptr(dz), L.stream())
        else:
            dz = dy
        M, O = dz.shape
        dx = gemm(dz, W) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1]:
            dW = torch.zeros_like(W)
            # reduction over the M samples: split K so that ~2048 workgroups exist (the output is only a few tiles)
            tiles = ((O + 127) // 128) * ((x2.shape[1] + 127) // 128)
            gemm(dz.t(), x2, out=dW, accumulate=True, split_k=max(1, min(M // 128, 2048 // tiles)))
        if ctx.needs_input_grad[2]:
            db = torch.zeros((O,), device=dz.device, dtype=torch.float32)
            L.call("moda_colsum_f32", L.ptr(dz), M, O, dz.stride(0), L.ptr(db), L.stream())
        return dx, dW, db, None


class EmbedFn(Function):
    """Embedding.forward (nerf.py:35-75), optionally on row-normalised input (rendering.py:64)."""

    @staticmethod
    def forward(ctx, x, n_freq, window, normalize):
        shape = x.shape
        xf = _f32(x).reshape(-1, shape[-1])
        C = shape[-1]
        out = torch.empty((xf.shape[0], C * (1 + 2 * n_freq)), device=xf.device, dtype=torch.float32)
        win = (L._F32 * 16)(*(list(window) + [0.0] * (16 - n_freq)))
        L.call("moda_embed_fwd", L.ptr(xf), xf.shape[0], C, n_freq, win, int(normalize), L.ptr(out), L.stream())
        ctx.save_for_backward(xf)
        ctx.meta = (shape, n_freq, list(window), int(normalize))
        return out.view(shape[:-1] + (out.shape[-1],))

    @staticmethod
    def backward(ctx, g):
        (xf,) = ctx.saved_tensors
        shape, n_freq, window, normalize = ctx.meta
        g2 = _f32(g).reshape(xf.shape[0], -1)
        dx = torch.empty_like(xf)
        win = (L._F32 * 16)(*(window + [0.0] * (16 - n_freq)))
        L.call("moda_embed_bwd", L.ptr(xf), xf.shape[0], shape[-1], n_freq, win, normalize, L.ptr(g2), L.ptr(dx),
               L.stream())
        return dx.view(shape), None, None, None


class PointsFn(Function):
    """xyz (N,S,3) = rays_o + rays_d * z (rendering.py:88-89)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, z):
        o, d, zz = _f32(rays_o), _f32(rays_d), _f32(z)
        N, S = zz.shape
        xyz = torch.empty((N, S, 3), device=zz.device, dtype=torch.float32)
        L.call("moda_points_fwd", L.ptr(o), L.ptr(d), L.ptr(zz), N, S, L.ptr(xyz), L.stream())
        ctx.save_for_backward(d, zz)
        return xyz

    @staticmethod
    def backward(ctx, g):
        d, zz = ctx.saved_tensors
        N, S = zz.shape
        g = _f32(g)
        do = torch.zeros((N, 3), device=g.device)
        dd = torch.zeros((N, 3), device=g.device)
        dz = torch.zeros((N, S), device=g.device) if ctx.needs_input_grad[2] else None
        L.call("moda_points_bwd", L.ptr(g), L.ptr(zz), L.ptr(d), N, S, L.ptr(do), L.ptr(dd), L.ptr(dz), L.stream())
        return do, dd, dz


class CompositeFn(Function):
    """inference() tail (rendering.py:183-237).  Returns rgb, feat_out, depth, sil, weights, visibility, vis_out, cyc_out."""

    @staticmethod
    def forward(ctx, rgbsigma, feat, z_vals, rays_d, beta, noise, xyz, clip, vis_pred, cyc):
        rs, z, rd, bt = _f32(rgbsigma), _f32(z_vals), _f32(rays_d), _f32(beta)
        N, S = z.shape
        dev = z.device
        ft = None if feat is None else _f32(feat)
        F = 0 if ft is None else ft.shape[-1]
        ns = None if noise is None else _f32(noise)
        xz = None if (xyz is None or clip is None) else _f32(xyz)
        cb = None if clip is None else _f32(clip)
        vp = None if vis_pred is None else _f32(vis_pred)
        cy = None if cyc is None else _f32(cyc)
        rgb = torch.empty((N, 3), device=dev)
        depth = torch.empty((N,), device=dev)
        sil = torch.empty((N,), device=dev)
        w = torch.empty((N, S), device=dev)
        vis = torch.empty((N, S), device=dev)
        fo = torch.empty((N, F), device=dev) if F else None
        vo = torch.empty((N,), device=dev) if vp is not None else None
        co = torch.empty((N,), device=dev) if cy is not None else None
        L.call("moda_composite_fwd", L.ptr(rs), L.ptr(ft), F, L.ptr(z), L.ptr(rd), L.ptr(bt), L.ptr(ns), L.ptr(xz), L.ptr(cb),
               L.ptr(vp), L.ptr(cy), N, S, L.ptr(rgb), L.ptr(fo), L.ptr(depth), L.ptr(sil), L.ptr(w), L.ptr(vis), L.ptr(vo),
               L.ptr(co), L.stream())
        ctx.save_for_backward(rs, ft, z, rd, bt, ns, xz, cb, vp, cy, w, vis)
        ctx.mark_non_differentiable(vis)
        if vo is not None:
            ctx.mark_non_differentiable(vo)
        return rgb, fo, depth, sil, w, vis, vo, co

    @staticmethod
    def backward(ctx, g_rgb, g_feat, g_depth, g_sil, g_w, _g_vis, _g_vo, g_cyc):
        rs, ft, z, rd, bt, ns, xz, cb, vp, cy, w, vis = ctx.saved_tensors
        N, S = z.shape
        dev = z.device
        F = 0 if ft is None else ft.shape[-1]
        c = lambda t: None if t is None else _f32(t)
        d_rs = torch.empty_like(rs)
        d_ft = torch.empty_like(ft) if (ft is not None and g_feat is not None) else None
        d_z = torch.zeros((N, S), device=dev)
        d_rd = torch.zeros((N, 3), device=dev)
        d_bt = torch.zeros((1,), device=dev)
        d_cy = torch.empty((N, S), device=dev) if (cy is not None and g_cyc is not None) else None
        L.call("moda_composite_bwd", L.ptr(rs), L.ptr(ft), F, L.ptr(z), L.ptr(rd), L.ptr(bt), L.ptr(ns), L.ptr(xz), L.ptr(cb),
               L.ptr(vp), L.ptr(cy), L.ptr(w), L.ptr(vis), N, S, L.ptr(c(g_rgb)), L.ptr(c(g_feat)), L.ptr(c(g_depth)),
               L.ptr(c(g_sil)), L.ptr(c(g_w)), L.ptr(c(g_cyc)), L.ptr(d_rs), L.ptr(d_ft), L.ptr(d_z), L.ptr(d_rd), L.ptr(d_bt),
               L.ptr(d_cy), L.stream())
        return d_rs, d_ft, d_z, d_rd, d_bt.view_as(bt), None, None, None, None, d_cy


class WarpFn(Function):
    """Skinning softmax + DQS blend / normalise / transform on prepared per-bone data (geom_utils.py:237-302, 457-517).
    prep (nsets,B,16) = [c | R | exp(scale) | 0], q (N,B,8) the dual quaternions blended as given."""

    @staticmethod
    def forward(ctx, prep, q, pts, dskin, skin_aux, cyc_ref):
        pr, qq, p, aux = _f32(prep), _f32(q), _f32(pts), _f32(skin_aux)
        N, S, _ = p.shape
        B = qq.shape[1]
        per_ray = 0 if pr.shape[0] == 1 else 1
        ds = None if dskin is None else _f32(dskin)
        cr = None if cyc_ref is None else _f32(cyc_ref)
        out = torch.empty_like(p)
        skin = torch.empty((N, S, B), device=p.device)
        cyc = torch.empty((N, S), device=p.device) if cr is not None else None
        L.call("moda_warp_prepped_fwd", L.ptr(pr), per_ray, L.ptr(qq), L.ptr(p), L.ptr(ds), 0, L.ptr(aux), N, S, B,
               L.ptr(out), L.ptr(skin), L.ptr(cr), L.ptr(cyc), L.stream())
        ctx.save_for_backward(pr, qq, p, skin, aux, cr)
        ctx.per_ray = per_ray
        ctx.has_dskin = ds is not None
        return out, cyc, skin

    @staticmethod
    def backward(ctx, g_out, g_cyc, g_skin):
        pr, qq, p, skin, aux, cr = ctx.saved_tensors
        N, S, _ = p.shape
        B = qq.shape[1]
        dev = p.device
        c = lambda t: None if t is None else _f32(t)
        d_p = torch.empty_like(p)
        d_ds = torch.empty((N, S, B), device=dev)
        d_pr_ray = torch.empty((N, B, 16), device=dev)
        d_q = torch.empty_like(qq)
        d_aux0 = torch.zeros((1,), device=dev)
        d_ref = torch.zeros_like(p) if cr is not None else None
        d_bl = torch.empty((N, S, 8), device=dev)
        L.call("moda_warp_prepped_bwd", L.ptr(pr), ctx.per_ray, L.ptr(qq), L.ptr(p), L.ptr(skin), L.ptr(aux), L.ptr(cr),
               L.ptr(c(g_out)), L.ptr(c(g_cyc) if cr is not None else None), L.ptr(c(g_skin)), N, S, B, L.ptr(d_p), L.ptr(d_ds),
               L.ptr(d_pr_ray), L.ptr(d_q), L.ptr(d_aux0), L.ptr(d_ref), L.ptr(d_bl), L.stream())
        if ctx.per_ray:
            d_pr = d_pr_ray
        else:   # shared rest bones: sum the per-ray partials over the rays
            d_pr = torch.zeros((B * 16,), device=dev)
            L.call("moda_colsum_f32", L.ptr(d_pr_ray), N, B * 16, B * 16, L.ptr(d_pr), L.stream())
            d_pr = d_pr.view(1, B, 16)
        d_aux = torch.zeros_like(aux)
        d_aux[0:1] = d_aux0
        return d_pr, d_q, d_p, (d_ds if ctx.has_dskin else None), d_aux, d_ref


class ProjectFn(Function):
    """obj_to_cam + pinhole_cam with the target view's rtk_vec (rendering.py:439-449)."""

    @staticmethod
    def forward(ctx, xyz, rtk_vec):
        x, r = _f32(xyz), _f32(rtk_vec)
        N, S, _ = x.shape
        out = torch.empty_like(x)
        L.call("moda_project_fwd", L.ptr(x), L.ptr(r), N, S, L.ptr(out), L.stream())
        ctx.save_for_backward(x, r)
        return out

    @staticmethod
    def backward(ctx, g):
        x, r = ctx.saved_tensors
        N, S, _ = x.shape
        dx = torch.empty_like(x)
        dr = torch.empty_like(r)
        L.call("moda_project_bwd", L.ptr(x), L.ptr(r), L.ptr(_f32(g)), N, S, L.ptr(dx), L.ptr(dr), L.stream())
        return dx, dr


class FlowRenderFn(Function):
    """vrender_flo (geom_utils.py:1704-1743) -> flo (N,2), valid (N,1)."""

    @staticmethod
    def forward(ctx, weights, proj, xys, img_size):
        w, p, xy = _f32(weights), _f32(proj), _f32(xys).reshape(-1, 2)
        N, S = w.shape
        flo = torch.empty((N, 2), device=w.device)
        valid = torch.empty((N, 1), device=w.device)
        L.call("moda_flow_render", L.ptr(w), L.ptr(p), L.ptr(xy), float(img_size), N, S, L.ptr(flo), L.ptr(valid), None, None,
               None, L.stream())
        ctx.save_for_backward(w, p, xy)
        ctx.img_size = float(img_size)
        ctx.mark_non_differentiable(valid)
        return flo, valid

    @staticmethod
    def backward(ctx, g_flo, _g_valid):
        w, p, xy = ctx.saved_tensors
        N, S = w.shape
        dw = torch.empty_like(w)
        dp = torch.empty_like(p)
        L.call("moda_flow_render", L.ptr(w), L.ptr(p), L.ptr(xy), ctx.img_size, N, S, None, None, L.ptr(_f32(g_flo)), L.ptr(dw),
               L.ptr(dp), L.stream())
        return dw, dp, None, None


class PtsExpFn(Function):
    """compute_pts_exp (loss_utils.py:165-175)."""

    @staticmethod
    def forward(ctx, weights, pts):
        w, p = _f32(weights), _f32(pts)
        N, S = w.shape
        out = torch.empty((N, 3), device=w.device)
        L.call("moda_pts_exp", L.ptr(w), L.ptr(p), N, S, L.ptr(out), None, None, None, L.stream())
        ctx.save_for_backward(w, p)
        return out

    @staticmethod
    def backward(ctx, g):
        w, p = ctx.saved_tensors
        N, S = w.shape
        dw = torch.empty_like(w)
        dp = torch.empty_like(p)
        L.call("moda_pts_exp", L.ptr(w), L.ptr(p), N, S, None, L.ptr(_f32(g)), L.ptr(dw), L.ptr(dp), L.stream())
        return dw, dp


# ---- per-(ray, bone) preparation, differentiable through torch ops on tiny tensors ------------------------------
def quaternion_to_matrix(q):
    r, i, j, k = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o


def _q_raw_mul(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def bone_prep(bones):
    """bones (nsets,B,10) -> (nsets,B,16): vec_to_sim3 (geom_utils.py:187-199) in the warp kernels' layout."""
    q = bones[..., 3:7]
    q = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    R = quaternion_to_matrix(q)
    pad = torch.zeros_like(bones[..., :1])
    return torch.cat([bones[..., :3], R, bones[..., 7:10].exp(), pad], -1).contiguous()


def bone_transform(bones, rts):
    """geom_utils.py:59-111 (neudbs): bones (B,10), rts (N,B,8) -> (N,B,10)."""
    N, B, _ = rts.shape
    dq_r, dq_d = rts[..., :4], rts[..., 4:]
    R = quaternion_to_matrix(dq_r).view(N, B, 3, 3)
    t = 2 * _q_raw_mul(dq_d, dq_r * dq_r.new_tensor([1, -1, -1, -1]))[..., 1:]
    center = (R @ bones[None, :, :3, None])[..., 0] + t
    orient = _q_raw_mul(dq_r, bones[None, :, 3:7].expand(N, B, 4))
    orient = torch.where(orient[..., :1] < 0, -orient, orient)
    return torch.cat([center, orient, bones[None, :, 7:10].expand(N, B, 3)], -1)


def dq_inverse(dq):
    """dual_quat.py:87-94"""
    return dq * dq.new_tensor([1, -1, -1, -1, 1, -1, -1, -1]) / (dq[..., :4] ** 2).sum(-1, keepdim=True)


# ---- loss heads behind compositing (rendering.py:410-437, 475-477, 573-578) ---------------------------------
class NormalizeFn(Function):
    """F.normalize(x, 2, -1)."""

    @staticmethod
    def forward(ctx, x):
        x2 = _f32(x).reshape(-1, x.shape[-1])
        y = torch.empty_like(x2)
        L.call("moda_normalize_rows", L.ptr(x2), x2.shape[0], x2.shape[1], L.ptr(y), None, None, L.stream())
        ctx.save_for_backward(x2)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        (x2,) = ctx.saved_tensors
        g2 = _f32(g).reshape(x2.shape)
        dx = torch.empty_like(x2)
        L.call("moda_normalize_rows", L.ptr(x2), x2.shape[0], x2.shape[1], None, L.ptr(g2), L.ptr(dx), L.stream())
        return dx.view(ctx.shape)


SINKHORN_ITERS = 20          # loss_utils.py:361
SINKHORN_TEMP = 0.03         # loss_utils.py:340


class FeatMatchFn(Function):
    """feat_match's cost volume -> matching probabilities -> expected grid location (loss_utils.py:326-389) on
    L2-normalised pixel features (N,16) and grid features (G,16).  use_ot: 20 Sinkhorn iterations on
    K = exp(-(1 - cost) / 0.03) (:338-374), row-normalised; else softmax(cost * kappa) with kappa = |beta| + 1e-9.
    The backward is the hand-derived reverse sweep through the iterations (no autograd tape of (N,G) matrices)."""

    @staticmethod
    def forward(ctx, feats_n, vol_n, query, kappa, use_ot):
        f, v, q, kp = _f32(feats_n), _f32(vol_n), _f32(query).reshape(-1, 3), _f32(kappa).reshape(1)
        N, G = f.shape[0], v.shape[0]
        dev = f.device
        Kmat = torch.empty((N, G), device=dev)
        L.call("moda_match_matrix", L.ptr(f), L.ptr(v), N, G, f.shape[1], L.ptr(kp), L.ptr(Kmat), L.stream())
        A = Bm = b = None
        if use_ot:
            T = SINKHORN_ITERS
            A = torch.empty((T + 1, N), device=dev)      # A[t] = a_t, a_0 = 1/N (:344-349)
            Bm = torch.empty((T, G), device=dev)         # Bm[t] = b_{t+1}
            A[0].fill_(1.0 / N)
            for t in range(T):
                L.call("moda_match_sweep", L.ptr(Kmat), N, G, 0, L.ptr(A[t]), 1, 1.0 / G, None, L.ptr(Bm[t]), L.stream())
                L.call("moda_match_sweep", L.ptr(Kmat), N, G, 1, L.ptr(Bm[t]), 1, 1.0 / N, None, L.ptr(A[t + 1]), L.stream())
            b = Bm[T - 1]
        pred = torch.empty((N, 3), device=dev)
        rowsum = torch.empty((N,), device=dev)
        L.call("moda_match_expect", L.ptr(Kmat), L.ptr(b), L.ptr(q), N, G, L.ptr(pred), L.ptr(rowsum), L.stream())
        ctx.save_for_backward(f, v, q, kp, Kmat, A, Bm, pred, rowsum)
        ctx.use_ot = bool(use_ot)
        return pred

    @staticmethod
    def backward(ctx, g_pred):
        f, v, q, kp, Kmat, A, Bm, pred, rowsum = ctx.saved_tensors
        N, G = Kmat.shape
        dev = Kmat.device
        gp = _f32(g_pred)
        Dbar = torch.empty_like(Kmat)
        kbar = None
        if ctx.use_ot:
            T = SINKHORN_ITERS
            Ubar = torch.empty((T, G), device=dev)       # Ubar[t] = ubar_{t+1}
            Wbar = torch.empty((T - 1, N), device=dev)   # Wbar[t] = wbar_{t+1}
            L.call("moda_match_ecols", L.ptr(Kmat), L.ptr(Bm[T - 1]), L.ptr(rowsum), L.ptr(gp), L.ptr(pred), L.ptr(q), N, G,
                   1.0 / G, L.ptr(Ubar[T - 1]), L.stream())
            for t in range(T, 1, -1):   # through u_t = K^T a_{t-1}, a_{t-1} = p1/(w_{t-1}+eps), w_{t-1} = K b_{t-1}, b_{t-1} = ...
                L.call("moda_match_sweep", L.ptr(Kmat), N, G, 1, L.ptr(Ubar[t - 1]), 2, 1.0 / N, L.ptr(A[t - 1]),
                       L.ptr(Wbar[t - 2]), L.stream())
                L.call("moda_match_sweep", L.ptr(Kmat), N, G, 0, L.ptr(Wbar[t - 2]), 2, 1.0 / G, L.ptr(Bm[t - 2]),
                       L.ptr(Ubar[t - 2]), L.stream())
            L.call("moda_match_dbar", L.ptr(Kmat), L.ptr(Bm[T - 1]), L.ptr(rowsum), L.ptr(gp), L.ptr(pred), L.ptr(q),
                   L.ptr(A), L.ptr(Ubar), T, L.ptr(Wbar), L.ptr(Bm), T - 1, N, G, L.ptr(kp), L.ptr(Dbar), None, L.stream())
        else:
            kbar = torch.zeros((1,), device=dev)
            L.call("moda_match_dbar", L.ptr(Kmat), None, L.ptr(rowsum), L.ptr(gp), L.ptr(pred), L.ptr(q), None, None, 0,
                   None, None, 0, N, G, L.ptr(kp), L.ptr(Dbar), L.ptr(kbar), L.stream())
        d_f = gemm(Dbar, v, out=torch.zeros_like(f), accumulate=True, split_k=max(1, min(32, G // 256)))
        d_v = gemm(Dbar.t(), f, out=torch.zeros_like(v), accumulate=True, split_k=max(1, min(8, N // 256)))
        return d_f, d_v, None, kbar, None


class LogSigLossFn(Function):
    """scale * sum_i -logsigmoid(sign * x_i) * (w_i | 1)  (visibility_loss, loss_utils.py:140,145) -> 0-dim tensor."""

    @staticmethod
    def forward(ctx, x, w, sign, scale):
        x1 = _f32(x).reshape(-1)
        w1 = None if w is None else _f32(w).reshape(-1)
        out = torch.zeros((1,), device=x1.device)
        L.call("moda_logsig_loss", L.ptr(x1), L.ptr(w1), x1.numel(), float(sign), float(scale), L.ptr(out), None, None,
               L.stream())
        ctx.save_for_backward(x1, w1)
        ctx.meta = (x.shape, float(sign), float(scale))
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        x1, w1 = ctx.saved_tensors
        shape, sign, scale = ctx.meta
        dx = torch.empty_like(x1)
        L.call("moda_logsig_loss", L.ptr(x1), L.ptr(w1), x1.numel(), sign, scale, None, L.ptr(_f32(g).reshape(1)), L.ptr(dx),
               L.stream())
        return dx.view(shape), None, None, None
