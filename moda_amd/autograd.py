"""torch.autograd.Functions over the HIP library: the training route of the rendering path.

The reference trains through PyTorch autograd on eager ops.  Here every arithmetic-heavy node is one Function whose
forward and backward are HIP kernels (exact fp32, MFMA GEMMs for the layers); torch's own autograd only stitches
them together through data-movement ops (cat / expand / slicing, whose backward is a copy or a segment sum) and the
per-(ray, bone) preparation of the skinning data (a few dozen flops on N*B elements, `bone_prep` / `bone_transform`).
Gradient parity is pinned against the reference's own autograd (tests/golden/g9_grad_*.npz).
"""
import os

import torch
from torch.autograd import Function

from . import _lib as L


def _f32(t):
    return L.dev(t)


def _dp(t):
    return None if t is None else t.data_ptr()


GEMM_BF16 = 1          # moda_hip.h MODA_GEMM_BF16
GEMM_BF16X3 = 32       # moda_hip.h MODA_GEMM_BF16X3
GEMM_BF16X6 = 64       # moda_hip.h MODA_GEMM_BF16X6
_TRAIN_PRECISION = os.environ.get("MODA_TRAIN_PRECISION", "fp32")      # exact fp32 unless the environment says otherwise (set_train_precision)
if _TRAIN_PRECISION not in ("fp32", "bf16", "bf16x3", "bf16x6"):
    raise ValueError(f"MODA_TRAIN_PRECISION={_TRAIN_PRECISION!r}: expected fp32, bf16, bf16x3 or bf16x6")
# bf16 training mode: run each network's forward as one launch of the fused PE+MLP kernel (activations dumped for the
# backward) instead of one GEMM per layer.  False keeps the per-layer GEMMs (A/B timing, tests).
FUSED_TRAIN_FORWARD = os.environ.get("MODA_FUSED_TRAIN_FORWARD", "1") != "0"
# with the fused forward: saved activations and the big backward tensors are held as bf16 (MODA_TRAIN_BF16_STORE, moda_hip.h)
TRAIN_BF16_STORE = os.environ.get("MODA_TRAIN_BF16_STORE", "1") != "0"
# multi-output Functions: let autograd hand None (instead of a freshly zero-filled tensor) for outputs nobody differentiates; every
# backward here and every kernel behind it takes a null gradient.  MODA_MATERIALIZE_GRADS=1 restores torch's default (an A/B switch)
MATERIALIZE_GRADS = os.environ.get("MODA_MATERIALIZE_GRADS", "0") == "1"
_STORE_FLAG = 2


def set_train_precision(mode):
    """Precision of the training route's GEMMs (every Linear of every network, forward and backward):
    'fp32' -- exact fp32 MFMA, the parity mode the gradient fixtures are checked in (default);
    'bf16' -- operands rounded to bf16 on their way into the MFMA, fp32 products, sums, master weights, activations and
    gradients (mixed precision as trainers usually run it): the throughput mode;
    'bf16x6' -- every fp32 operand split EXACTLY into three bf16 (hi + mid + lo) on its way into the MFMA, six bf16 MFMAs per
    product: the accuracy class of 'fp32' (held to the same bars) on HBM-bound kernels -- the fast parity mode;
    'bf16x3' -- two bf16 per operand (16 significand bits), three MFMAs: results within ~1e-6 of 'fp32'; a ReLU whose
    pre-activation is that close to zero may switch, so single samples' gradients differ sparsely.
    In both everything stays fp32 in memory and the per-layer forward is used (the fused bf16-storage forward belongs to 'bf16')."""
    global _TRAIN_PRECISION
    if mode not in ("fp32", "bf16", "bf16x3", "bf16x6"):
        raise ValueError(mode)
    _TRAIN_PRECISION = mode


def get_train_precision():
    return _TRAIN_PRECISION


def _gemm_flags():
    return {"bf16": GEMM_BF16, "bf16x3": GEMM_BF16X3, "bf16x6": GEMM_BF16X6}.get(_TRAIN_PRECISION, 0)



class _ZeroPool:
    """Small zero-initialised gradient buffers as exclusive slices of one pre-zeroed block: a backward pass asks for ~30 of them
    (targets of atomic accumulation, a few floats to a megabyte each), and each torch.zeros is a fill launch of its own.  A
    block is never handed out twice -- when it is used up the next request allocates (and zero-fills) a new one, the old block
    lives as long as any slice of it does -- so a slice behaves exactly like a fresh torch.zeros.  Inside a stream capture the
    block must be allocated (and its fill recorded) by that capture, or a replay would accumulate into the previous replay's
    sums: every capture (keyed by the runtime's capture id) starts a block of its own.  MODA_ZERO_POOL=0 restores one fill per
    buffer."""
    CAP = 1 << 22                      # floats per block (16 MB: one ~5 us fill)
    ON = os.environ.get("MODA_ZERO_POOL", "1") != "0"

    def __init__(self):
        self.blocks = {}               # stream -> [buf, off, key]

    def get(self, shape, device):
        shape = tuple(int(v) for v in shape)
        n = 1
        for v in shape:
            n *= v
        span = (n + 63) // 64 * 64                                   # 256-byte aligned slices (vector loads, atomics)
        if not self.ON or n == 0 or span > self.CAP // 8 or torch.device(device).type != "cuda":
            return torch.zeros(shape, device=device, dtype=torch.float32)
        # One block per STREAM: a block is allocated and zero-filled on the stream that asks first, and a slice handed
        # to work on another stream would have no ordering against that fill (nor would the caching allocator know the other
        # stream still writes to it) -- the torch.zeros this replaces is stream-safe per call.  Key besides: the device and the
        # runtime's capture id (0 outside a capture: two captures in a row must not share a block either).
        st = torch.cuda.current_stream(device).cuda_stream
        key = (torch.device(device), int(L.load().moda_stream_capture_id(L.stream())))
        blk = self.blocks.get(st)
        if blk is None or blk[2] != key or blk[1] + span > self.CAP:
            blk = [torch.zeros((self.CAP,), device=device, dtype=torch.float32), 0, key]
            self.blocks[st] = blk
        v = blk[0][blk[1]:blk[1] + n].view(shape)
        blk[1] += span
        return v


_ZEROS = _ZeroPool()


def zeros(shape, device):
    """A zero fp32 tensor of `shape` for a kernel to accumulate into (see _ZeroPool)."""
    return _ZEROS.get(shape, device)


def zeros_like(t):
    return _ZEROS.get(t.shape, t.device)


def gemm(a, b, bias=None, act=0, mask_src=None, out=None, accumulate=False, split_k=1, a2=None, rowbias=None,
         rows_per_bias=1, exact=False):
    """out (M,N) = epi(a @ b [+ a2 @ b[K1:]]) for fp32 CUDA matrices (views are fine).

    a (M,K1) [, a2 (M,K-K1): the reduction runs over cat([a, a2], 1) without materialising it], b (K,N);
    bias (N) per column, rowbias (ceil(M/rows_per_bias), N) per group of rows; act 0 none / 1 relu / 2 sigmoid;
    mask_src (M,N): result zeroed where mask_src <= 0; accumulate False | True / 1 (atomic +=, split_k allowed) |
    2 (out = epi(out + result)).  Operands with one unit stride go to the pipelined kernel, anything else to the
    generic strided one.  exact=True: exact fp32 whatever the training precision (the inference route's weight products)."""
    M, K1 = a.shape
    K, N = b.shape
    assert K == K1 + (0 if a2 is None else a2.shape[1])
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    acc = int(accumulate)
    a_ok = a.stride(1) == 1 or (a.stride(0) == 1 and a2 is None) or M == 1 or K1 == 1
    b_ok = b.stride(0) == 1 or b.stride(1) == 1 or K == 1 or N == 1
    if a_ok and b_ok and out.stride(1) == 1 and (a2 is None or a2.stride(1) == 1) \
            and (mask_src is None or mask_src.stride(1) == 1) and (rowbias is None or rowbias.stride(1) == 1):
        sam, sak = a.stride(0), a.stride(1)
        if K1 == 1 and sak != 1:
            sak = 1          # a single column: the k stride is never used
        elif M == 1 and sak != 1:
            sam = 1
        sbk, sbn = b.stride(0), b.stride(1)
        if N == 1 and sbk != 1 and sbn != 1:
            sbn = 1
        elif K == 1 and sbk != 1 and sbn != 1:
            sbk = 1
        d = L.GemmDesc(A=_dp(a), sam=sam, sak=sak, A2=_dp(a2), sam2=0 if a2 is None else a2.stride(0), K1=K1,
                       B=_dp(b), sbk=sbk, sbn=sbn, C=_dp(out), ldc=out.stride(0), M=M, N=N, K=K, bias=_dp(bias),
                       rowbias=_dp(rowbias), ld_rowbias=0 if rowbias is None else rowbias.stride(0),
                       rows_per_bias=int(rows_per_bias), mask_src=_dp(mask_src),
                       ld_mask=0 if mask_src is None else mask_src.stride(0), act=int(act), accumulate=acc,
                       split_k=int(split_k), reserved=0 if exact else _gemm_flags(), a_sum=None, mask_bits=None, ld_bits=0)
        L.call("moda_gemm_f32_ex", L._c.byref(d), L.stream())
        return out
    if a2 is not None or rowbias is not None or acc == 2 or (mask_src is not None and mask_src.stride(0) != out.stride(0)):
        raise NotImplementedError("gemm: split / row-bias / += forms need unit-stride operands")
    L.call("moda_gemm_f32", L.ptr(a), a.stride(0), a.stride(1), L.ptr(b), b.stride(0), b.stride(1), L.ptr(out),
           out.stride(0), M, N, K, L.ptr(bias), act, L.ptr(mask_src), acc, int(split_k), L.stream())
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
        y = gemm(x2, Wc.t(), bias=None if b is None else _f32(b), act=act)
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
            dW = zeros_like(W)
            # reduction over the M samples: split K so that ~1024 workgroups exist (the output is only a few tiles)
            tiles = ((O + 127) // 128) * ((x2.shape[1] + 127) // 128)
            gemm(dz.t(), x2, out=dW, accumulate=True, split_k=max(1, min(M // 256, 1024 // tiles)))
        if ctx.needs_input_grad[2]:
            db = zeros((O,), dz.device)
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
        L.call("moda_embed_fwd", L.ptr(xf), xf.shape[0], C, n_freq, win, int(normalize), L.ptr(out), out.stride(0), L.stream())
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
        L.call("moda_embed_bwd", L.ptr(xf), xf.shape[0], shape[-1], n_freq, win, normalize, L.ptr(g2), g2.stride(0), L.ptr(dx),
               L.stream())
        return dx.view(shape), None, None, None


class EmbedJacTFn(Function):
    """out (P,C) = J(x)^T g for the encoding at CONSTANT positions x (P,C), g (P, C (1 + 2 n_freq)): the last step of the eikonal
    term's analytic d sigma / d x (loss_utils.py:20-46).  Differentiable w.r.t. g (its backward is the encoding's tangent,
    moda_embed_jvp), which is what the double backward of the reference differentiates."""

    @staticmethod
    def forward(ctx, x, g, n_freq, window):
        xf, g2 = _f32(x), _f32(g)
        out = torch.empty_like(xf)
        win = (L._F32 * 16)(*(list(window) + [0.0] * (16 - n_freq)))
        L.call("moda_embed_bwd", L.ptr(xf), xf.shape[0], xf.shape[1], n_freq, win, 0, L.ptr(g2), g2.stride(0), L.ptr(out), L.stream())
        ctx.save_for_backward(xf)
        ctx.meta = (n_freq, list(window), g2.shape[1])
        return out

    @staticmethod
    def backward(ctx, u):
        (xf,) = ctx.saved_tensors
        n_freq, window, width = ctx.meta
        u2 = _f32(u)
        dg = torch.zeros((xf.shape[0], width), device=xf.device, dtype=torch.float32) if width > xf.shape[1] * (1 + 2 * n_freq) \
            else torch.empty((xf.shape[0], width), device=xf.device, dtype=torch.float32)
        win = (L._F32 * 16)(*(window + [0.0] * (16 - n_freq)))
        L.call("moda_embed_jvp", L.ptr(xf), xf.shape[0], xf.shape[1], n_freq, win, L.ptr(u2), L.ptr(dg), dg.stride(0), L.stream())
        return None, dg, None, None


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
        do = zeros((N, 3), g.device)
        dd = zeros((N, 3), g.device)
        dz = zeros((N, S), g.device) if ctx.needs_input_grad[2] else None
        L.call("moda_points_bwd", L.ptr(g), L.ptr(zz), L.ptr(d), N, S, L.ptr(do), L.ptr(dd), L.ptr(dz), L.stream())
        return do, dd, dz


class CompositeFn(Function):
    """inference() tail (rendering.py:183-237).  Returns rgb, feat_out, depth, sil, weights, visibility, vis_out, cyc_out."""

    @staticmethod
    def forward(ctx, rgbsigma, feat, z_vals, rays_d, beta, noise, xyz, clip, vis_pred, cyc, rgb_filter_scale=0.0):
        ctx.set_materialize_grads(MATERIALIZE_GRADS)   # off: an output nobody differentiates arrives as None, not as a zero fill
        rs, z, rd, bt = _f32(rgbsigma), _f32(z_vals), _f32(rays_d), _f32(beta)
        ctx.rgb_filter_scale = float(rgb_filter_scale)
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
               L.ptr(vp), L.ptr(cy), ctx.rgb_filter_scale, N, S, L.ptr(rgb), L.ptr(fo), L.ptr(depth), L.ptr(sil), L.ptr(w),
               L.ptr(vis), L.ptr(vo), L.ptr(co), None, 0.0, None, L.stream())     # no early termination: every sample composes
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
        d_z = zeros((N, S), dev)
        d_rd = zeros((N, 3), dev)
        d_bt = zeros((1,), dev)
        d_cy = torch.empty((N, S), device=dev) if (cy is not None and g_cyc is not None) else None
        L.call("moda_composite_bwd", L.ptr(rs), L.ptr(ft), F, L.ptr(z), L.ptr(rd), L.ptr(bt), L.ptr(ns), L.ptr(xz), L.ptr(cb),
               L.ptr(vp), L.ptr(cy), L.ptr(w), L.ptr(vis), ctx.rgb_filter_scale, N, S, L.ptr(c(g_rgb)), L.ptr(c(g_feat)),
               L.ptr(c(g_depth)),
               L.ptr(c(g_sil)), L.ptr(c(g_w)), L.ptr(c(g_cyc)), L.ptr(d_rs), L.ptr(d_ft), L.ptr(d_z), L.ptr(d_rd), L.ptr(d_bt),
               L.ptr(d_cy), L.stream())
        return d_rs, d_ft, d_z, d_rd, d_bt.view_as(bt), None, None, None, None, d_cy, None


class WarpFn(Function):
    """Skinning softmax + DQS blend / normalise / transform on prepared per-bone data (geom_utils.py:237-302, 457-517).
    prep (nsets,B,16) = [c | R | exp(scale) | 0], q (N,B,8) the dual quaternions blended as given.  pts_tf (N,S,3)|None:
    the points the blended transform is applied to when they are not the points the weights are evaluated at
    (x + nerf_dis(x), geom_utils.py:420-425)."""

    @staticmethod
    def forward(ctx, prep, q, pts, dskin, skin_aux, cyc_ref, pts_tf=None):
        ctx.set_materialize_grads(MATERIALIZE_GRADS)   # off: an output nobody differentiates arrives as None, not as a zero fill
        pr, qq, p, aux = _f32(prep), _f32(q), _f32(pts), _f32(skin_aux)
        N, S, _ = p.shape
        B = qq.shape[1]
        per_ray = 0 if pr.shape[0] == 1 else 1
        ds = None if dskin is None else _f32(dskin)
        cr = None if cyc_ref is None else _f32(cyc_ref)
        pt = None if pts_tf is None else _f32(pts_tf)
        out = torch.empty_like(p)
        skin = torch.empty((N, S, B), device=p.device)
        cyc = torch.empty((N, S), device=p.device) if cr is not None else None
        L.call("moda_warp_prepped_fwd", L.ptr(pr), per_ray, L.ptr(qq), L.ptr(p), L.ptr(pt), L.ptr(ds), 0, L.ptr(aux), N, S, B,
               L.ptr(out), L.ptr(skin), L.ptr(cr), L.ptr(cyc), L.stream())
        ctx.save_for_backward(pr, qq, p, skin, aux, cr, pt)
        ctx.per_ray = per_ray
        ctx.has_dskin = ds is not None
        return out, cyc, skin

    @staticmethod
    def backward(ctx, g_out, g_cyc, g_skin):
        pr, qq, p, skin, aux, cr, pt = ctx.saved_tensors
        N, S, _ = p.shape
        B = qq.shape[1]
        dev = p.device
        c = lambda t: None if t is None else _f32(t)
        d_p = torch.empty_like(p)
        d_pt = torch.empty_like(p) if pt is not None else None
        d_ds = torch.empty((N, S, B), device=dev)
        d_pr_ray = torch.empty((N, B, 16), device=dev)
        d_q = torch.empty_like(qq)
        d_aux = zeros_like(aux)                  # the kernel accumulates d skin_aux[0] into its first element; [1] has no gradient
        d_aux0 = d_aux
        d_ref = zeros_like(p) if cr is not None else None
        d_bl = torch.empty((N, S, 8), device=dev)
        L.call("moda_warp_prepped_bwd", L.ptr(pr), ctx.per_ray, L.ptr(qq), L.ptr(p), L.ptr(pt), L.ptr(d_pt), L.ptr(skin),
               L.ptr(aux), L.ptr(cr), L.ptr(c(g_out)), L.ptr(c(g_cyc) if cr is not None else None), L.ptr(c(g_skin)), N, S, B,
               L.ptr(d_p), L.ptr(d_ds), L.ptr(d_pr_ray), L.ptr(d_q), L.ptr(d_aux0), L.ptr(d_ref), L.ptr(d_bl), L.stream())
        if ctx.per_ray:
            d_pr = d_pr_ray
        else:   # shared rest bones: sum the per-ray partials over the rays
            d_pr = zeros((B * 16,), dev)
            L.call("moda_colsum_f32", L.ptr(d_pr_ray), N, B * 16, B * 16, L.ptr(d_pr), L.stream())
            d_pr = d_pr.view(1, B, 16)
        return d_pr, d_q, d_p, (d_ds if ctx.has_dskin else None), d_aux, d_ref, d_pt


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


# ---- per-(ray, bone) preparation: one kernel forward, one backward each ------------------------------------------
class BonePrepFn(Function):
    """bones (nsets,B,10) -> (nsets,B,16): vec_to_sim3 (geom_utils.py:187-199) in the warp kernels' layout."""

    @staticmethod
    def forward(ctx, bones):
        b = _f32(bones)
        n = b.numel() // 10
        out = torch.empty(b.shape[:-1] + (16,), device=b.device)
        L.call("moda_bone_prep", L.ptr(b), n, L.ptr(out), None, None, L.stream())
        ctx.save_for_backward(b)
        return out

    @staticmethod
    def backward(ctx, g):
        (b,) = ctx.saved_tensors
        d = torch.empty_like(b)
        L.call("moda_bone_prep", L.ptr(b), b.numel() // 10, None, L.ptr(_f32(g)), L.ptr(d), L.stream())
        return d


class BoneTransformFn(Function):
    """geom_utils.py:59-111 (neudbs): bones (B,10), rts (N,B,8) -> (N,B,10)."""

    @staticmethod
    def forward(ctx, bones, rts):
        b, r = _f32(bones), _f32(rts)
        N, B, _ = r.shape
        out = torch.empty((N, B, 10), device=r.device)
        L.call("moda_bone_transform_fwd", L.ptr(b), L.ptr(r), N, B, L.ptr(out), None, L.stream())
        ctx.save_for_backward(b, r)
        return out

    @staticmethod
    def backward(ctx, g):
        b, r = ctx.saved_tensors
        N, B, _ = r.shape
        d_ray = torch.empty((N, B, 10), device=r.device)
        d_r = torch.empty_like(r)
        L.call("moda_bone_transform_bwd", L.ptr(b), L.ptr(r), N, B, L.ptr(_f32(g)), L.ptr(d_ray), L.ptr(d_r), L.stream())
        d_b = zeros((B * 10,), r.device)
        L.call("moda_colsum_f32", L.ptr(d_ray), N, B * 10, B * 10, L.ptr(d_b), L.stream())
        return d_b.view(B, 10), d_r


class DqInverseFn(Function):
    """dual_quat.py:87-94"""

    @staticmethod
    def forward(ctx, dq):
        q = _f32(dq)
        out = torch.empty_like(q)
        L.call("moda_dq_op", 5, L.ptr(q), None, q.numel() // 8, L.ptr(out), None, L.stream())
        ctx.save_for_backward(q)
        return out

    @staticmethod
    def backward(ctx, g):
        (q,) = ctx.saved_tensors
        d = torch.empty_like(q)
        L.call("moda_dq_inverse_bwd", L.ptr(q), L.ptr(_f32(g)), q.numel() // 8, L.ptr(d), L.stream())
        return d


def bone_prep(bones):
    return BonePrepFn.apply(bones)


def bone_transform(bones, rts):
    return BoneTransformFn.apply(bones, rts)


def dq_inverse(dq):
    return DqInverseFn.apply(dq)


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
# throughput mode of the training route: the matching matrix and its transpose are held as bf16 (its 78 sweeps per step read
# them 78 times); every vector, sum and result stays fp32.  MODA_MATCH_BF16=0 keeps them fp32 in that mode too.
MATCH_BF16 = os.environ.get("MODA_MATCH_BF16", "1") != "0"
SINKHORN_TEMP = 0.03         # loss_utils.py:340
# the 40 + 38 sweeps of the Sinkhorn iterations as ONE persistent launch each way (moda_match_sinkhorn: bf16 matrix, N % 512 == 0,
# N <= 2048; the matrix stays in LDS / registers across the sweeps, a flag-array grid barrier between them).  Built, correct
# (1e-7 of the per-sweep chain), and OPT-IN (MODA_SINKHORN_PERSIST=1): measured on one MI355X at cfg4's size it costs 6.3 us per
# sweep -- 2.7 us the cross-XCD barrier (an sc1 flag store and an sc1 poll are two fabric round trips), 2.7 us the arithmetic of
# 65 M bf16 multiply-adds on 8 waves per CU (unpacking bf16 is half of its VALU work), 0.4 us staging the vector, + 100 us per
# step to load the matrices -- against 7.0 us for a sweep as its own launch reading the matrix from the Infinity Cache: the
# captured step is 0.05 ms SLOWER with it, the eagerly launched one 0.35 ms faster (tools/sinkhorn_bench.py).
SINKHORN_PERSIST = os.environ.get("MODA_SINKHORN_PERSIST", "0") == "1"
_SINK_FLAGS = 320            # int32 words handed to the kernel: one per workgroup (CU) + the time-out word


def _sinkhorn_persistent(Kmat, KmatT, N, G, T, backward, A, Bm, Ubar, Wbar):
    """True when the persistent kernel took the chain; False: the caller runs the per-sweep launches."""
    if not SINKHORN_PERSIST or Kmat.dtype != torch.bfloat16 or N % 512 or N > 2048 or G % 8:
        return False
    # its grid barrier needs every workgroup resident at once: never beside another process on the same GPU (the ranks-on-one-GPU
    # test harness).  A barrier that does time out poisons the chain's last vectors with NaN (loss_kernels.hip): loud, not silent.
    if os.environ.get("MODA_BENCH_ONE_GPU") == "1":
        return False
    flags = zeros((_SINK_FLAGS,), Kmat.device)               # (zero words: a pre-zeroed pool slice, re-zeroed by a replayed graph)
    rc = L.load().moda_match_sinkhorn(L.ptr(Kmat), L.ptr(KmatT), N, G, T, int(backward), L.ptr(A), L.ptr(Bm), L.ptr(Ubar),
                                      L.ptr(Wbar), L.ptr(flags), _SINK_FLAGS, L.stream())
    if rc == -2:                                             # MODA_ESHAPE: not served on this device / shape
        return False
    if rc != 0:
        raise RuntimeError(f"moda_match_sinkhorn failed with code {rc}")
    return True


class FeatMatchFn(Function):
    """feat_match's cost volume -> matching probabilities -> expected grid location (loss_utils.py:326-389) on
    L2-normalised pixel features (N,16) and grid features (G,16).  use_ot: 20 Sinkhorn iterations on
    K = exp(-(1 - cost) / 0.03) (:338-374), row-normalised; else softmax(cost * kappa) with kappa = |beta| + 1e-9.
    The backward is the hand-derived reverse sweep through the iterations (no autograd tape of (N,G) matrices)."""

    @staticmethod
    def forward(ctx, feats_n, vol_n, query, kappa, use_ot, want_prob=False):
        ctx.set_materialize_grads(MATERIALIZE_GRADS)   # off: an output nobody differentiates arrives as None, not as a zero fill
        f, v, q, kp = _f32(feats_n), _f32(vol_n), _f32(query).reshape(-1, 3), _f32(kappa).reshape(1)
        N, G = f.shape[0], v.shape[0]
        dev = f.device
        kb = int(_TRAIN_PRECISION == "bf16" and MATCH_BF16)
        kdt = torch.bfloat16 if kb else torch.float32
        Kmat = torch.empty((N, G), device=dev, dtype=kdt)
        L.call("moda_match_matrix", L.ptr(f), L.ptr(v), N, G, f.shape[1], L.ptr(kp), L.ptr(Kmat), kb, L.stream())
        A = Bm = b = KmatT = None
        if use_ot:
            KmatT = torch.empty((G, N), device=dev, dtype=kdt)      # same entries, pixels along the rows' fast axis
            L.call("moda_match_matrix", L.ptr(v), L.ptr(f), G, N, f.shape[1], L.ptr(kp), L.ptr(KmatT), kb, L.stream())
            T = SINKHORN_ITERS
            A = torch.empty((T + 1, N), device=dev)      # A[t] = a_t, a_0 = 1/N (:344-349)
            Bm = torch.empty((T, G), device=dev)         # Bm[t] = b_{t+1}
            A[0].fill_(1.0 / N)
            if not _sinkhorn_persistent(Kmat, KmatT, N, G, T, False, A, Bm, None, None):
                for t in range(T):
                    L.call("moda_match_sweep", L.ptr(KmatT), G, N, L.ptr(A[t]), 1, 1.0 / G, None, L.ptr(Bm[t]), kb, L.stream())
                    L.call("moda_match_sweep", L.ptr(Kmat), N, G, L.ptr(Bm[t]), 1, 1.0 / N, None, L.ptr(A[t + 1]), kb, L.stream())
            b = Bm[T - 1]
        pred = torch.empty((N, 3), device=dev)
        rowsum = torch.empty((N,), device=dev)
        L.call("moda_match_expect", L.ptr(Kmat), L.ptr(b), L.ptr(q), N, G, L.ptr(pred), L.ptr(rowsum), kb, L.stream())
        prob = None
        if want_prob:     # the matching probabilities themselves, for the back-correspondence term (loss_utils.py:386-391)
            prob = torch.empty((N, G), device=dev)
            L.call("moda_match_prob", L.ptr(Kmat), L.ptr(b), L.ptr(rowsum), N, G, L.ptr(prob), kb, L.stream())
        ctx.save_for_backward(f, v, q, kp, Kmat, KmatT, A, Bm, pred, rowsum, prob)
        ctx.use_ot = bool(use_ot)
        ctx.kb = kb
        return pred, prob

    @staticmethod
    def backward(ctx, g_pred, g_prob):
        f, v, q, kp, Kmat, KmatT, A, Bm, pred, rowsum, prob = ctx.saved_tensors
        N, G = Kmat.shape
        dev = Kmat.device
        kb = ctx.kb
        gp = zeros((N, 3), dev) if g_pred is None else _f32(g_pred)
        gP = gPT = sP = None
        if g_prob is not None and prob is not None:
            gP = _f32(g_prob)
            sP = (gP * prob).sum(1).contiguous()         # sum_g g_prob prob per pixel
            gPT = gP.t().contiguous() if ctx.use_ot else None
        Dbar = torch.empty((N, G), device=dev, dtype=torch.float32)
        kbar = None
        if ctx.use_ot:
            T = SINKHORN_ITERS
            Ubar = torch.empty((T, G), device=dev)       # Ubar[t] = ubar_{t+1}
            Wbar = torch.empty((T - 1, N), device=dev)   # Wbar[t] = wbar_{t+1}
            L.call("moda_match_ecols", L.ptr(KmatT), L.ptr(Bm[T - 1]), L.ptr(rowsum), L.ptr(gp), L.ptr(pred), L.ptr(q),
                   L.ptr(gPT), L.ptr(sP), N, G, 1.0 / G, L.ptr(Ubar[T - 1]), kb, L.stream())
            if not _sinkhorn_persistent(Kmat, KmatT, N, G, T, True, A, Bm, Ubar, Wbar):
                for t in range(T, 1, -1):   # through u_t = K^T a_{t-1}, a_{t-1} = p1/(w_{t-1}+eps), w_{t-1} = K b_{t-1}, b_{t-1} = ...
                    L.call("moda_match_sweep", L.ptr(Kmat), N, G, L.ptr(Ubar[t - 1]), 2, 1.0 / N, L.ptr(A[t - 1]),
                           L.ptr(Wbar[t - 2]), kb, L.stream())
                    L.call("moda_match_sweep", L.ptr(KmatT), G, N, L.ptr(Wbar[t - 2]), 2, 1.0 / G, L.ptr(Bm[t - 2]),
                           L.ptr(Ubar[t - 2]), kb, L.stream())
            L.call("moda_match_dbar", L.ptr(Kmat), L.ptr(Bm[T - 1]), L.ptr(rowsum), L.ptr(gp), L.ptr(pred), L.ptr(q),
                   L.ptr(A), L.ptr(Ubar), T, L.ptr(Wbar), L.ptr(Bm), T - 1, L.ptr(gP), L.ptr(sP), N, G, L.ptr(kp), L.ptr(Dbar),
                   None, kb, L.stream())
        else:
            kbar = zeros((1,), dev)
            L.call("moda_match_dbar", L.ptr(Kmat), None, L.ptr(rowsum), L.ptr(gp), L.ptr(pred), L.ptr(q), None, None, 0,
                   None, None, 0, L.ptr(gP), L.ptr(sP), N, G, L.ptr(kp), L.ptr(Dbar), L.ptr(kbar), kb, L.stream())
        d_f = gemm(Dbar, v, out=zeros_like(f), accumulate=True, split_k=max(1, min(32, G // 256)))
        d_v = gemm(Dbar.t(), f, out=zeros_like(v), accumulate=True, split_k=max(1, min(8, N // 256)))
        return d_f, d_v, None, kbar, None, None


class RayLossFn(Function):
    """The img / sil / flo terms of inference_deform with their batch statistics (rendering.py:518-571) as one kernel each way.
    -> img_loss_samp (N,1), sil_loss_samp (N,1), flo_loss_samp (N,1), sil_at_samp_flo (N,1) bool."""

    @staticmethod
    def forward(ctx, rgb, sil, flo, valid, img_at, sil_at, vis_at, flo_at, cfd_at, training):
        ctx.set_materialize_grads(MATERIALIZE_GRADS)   # off: an output nobody differentiates arrives as None, not as a zero fill
        c = lambda t, n: _f32(t).reshape(-1, n) if n > 1 else _f32(t).reshape(-1)
        rgb_, sil_, flo_, val_ = c(rgb, 3), c(sil, 1), c(flo, 2), c(valid, 1)
        ia, sa, va, fa, ca = c(img_at, 3), c(sil_at, 1), c(vis_at, 1), c(flo_at, 2), c(cfd_at, 1)
        N = sil_.shape[0]
        dev = sil_.device
        out = torch.empty((3, N), device=dev)
        sflo = torch.empty((N,), device=dev, dtype=torch.uint8)
        stats = torch.empty((8,), device=dev)
        L.call("moda_ray_loss", L.ptr(rgb_), L.ptr(sil_), L.ptr(flo_), L.ptr(val_), L.ptr(ia), L.ptr(sa), L.ptr(va), L.ptr(fa),
               L.ptr(ca), N, int(bool(training)), L.ptr(out[0]), L.ptr(out[1]), L.ptr(out[2]), L.ptr(sflo), L.ptr(stats),
               None, None, None, None, None, None, L.stream())
        ctx.save_for_backward(rgb_, sil_, flo_, ia, sa, va, fa, ca, stats)
        ctx.training = int(bool(training))
        ctx.shapes = (rgb.shape, sil.shape, flo.shape)
        sf = sflo.view(N, 1).bool()
        ctx.mark_non_differentiable(sf)
        return out[0].view(N, 1), out[1].view(N, 1), out[2].view(N, 1), sf

    @staticmethod
    def backward(ctx, g_img, g_sil, g_flo, _g):
        rgb_, sil_, flo_, ia, sa, va, fa, ca, stats = ctx.saved_tensors
        N = sil_.shape[0]
        c = lambda t: None if t is None else _f32(t).reshape(-1)
        d_rgb, d_sil, d_flo = torch.empty_like(rgb_), torch.empty_like(sil_), torch.empty_like(flo_)
        L.call("moda_ray_loss", L.ptr(rgb_), L.ptr(sil_), L.ptr(flo_), None, L.ptr(ia), L.ptr(sa), L.ptr(va), L.ptr(fa), L.ptr(ca),
               N, ctx.training, None, None, None, None, L.ptr(stats), L.ptr(c(g_img)), L.ptr(c(g_sil)), L.ptr(c(g_flo)),
               L.ptr(d_rgb), L.ptr(d_sil), L.ptr(d_flo), L.stream())
        sh = ctx.shapes
        return d_rgb.view(sh[0]), d_sil.view(sh[1]), d_flo.view(sh[2]), None, None, None, None, None, None, None


class MaskedMeanFn(Function):
    """x[mask].mean() without the boolean gather: x (N, k) or (N,), mask (N, 1) / (N,) (non-zero = selected) -> 0-dim."""

    @staticmethod
    def forward(ctx, x, mask):
        x2 = _f32(x)
        N = x2.shape[0]
        x2 = x2.reshape(N, -1)
        m = mask.reshape(N).to(torch.float32).contiguous()
        out = torch.empty((2,), device=x2.device)
        L.call("moda_masked_mean", L.ptr(x2), L.ptr(m), N, x2.shape[1], L.ptr(out), None, None, L.stream())
        ctx.save_for_backward(m, out)
        ctx.shape = x.shape
        ctx.k = x2.shape[1]
        return out[0]

    @staticmethod
    def backward(ctx, g):
        m, out = ctx.saved_tensors
        N = m.shape[0]
        dx = torch.empty((N, ctx.k), device=m.device)
        L.call("moda_masked_mean", None, L.ptr(m), N, ctx.k, L.ptr(out), L.ptr(_f32(g).reshape(1)), L.ptr(dx), L.stream())
        return dx.view(ctx.shape), None


class RowDistFn(Function):
    """Per-row distance of a and b (..., F): ||a - b||_2 (loss_utils.py:200, :216-221) or, mean_sq=True, mean_c (a - b)^2
    (rendering.py:573-577) -> (...,).  One kernel each way (moda_row_dist) where the eager form is 3 + ~9 launches."""

    @staticmethod
    def forward(ctx, a, b, mean_sq=False):
        F = a.shape[-1]
        a2, b2 = _f32(a).reshape(-1, F), _f32(b).expand(a.shape).reshape(-1, F)
        if not b2.is_contiguous():
            b2 = b2.contiguous()
        out = torch.empty((a2.shape[0],), device=a2.device)
        L.call("moda_row_dist", L.ptr(a2), L.ptr(b2), a2.shape[0], F, int(mean_sq), L.ptr(out), None, None, None, L.stream())
        ctx.save_for_backward(a2, b2)
        ctx.meta = (tuple(a.shape), tuple(b.shape), int(mean_sq))
        return out.view(a.shape[:-1])

    @staticmethod
    def backward(ctx, g):
        a2, b2 = ctx.saved_tensors
        sa, sb, mean_sq = ctx.meta
        na, nb = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if nb and sb != sa:
            raise NotImplementedError("RowDistFn: gradient towards a broadcast b")
        da = torch.empty_like(a2) if na else None
        db = torch.empty_like(b2) if nb else None
        if na or nb:
            L.call("moda_row_dist", L.ptr(a2), L.ptr(b2), a2.shape[0], a2.shape[1], mean_sq, None, L.ptr(_f32(g).reshape(-1)),
                   L.ptr(da), L.ptr(db), L.stream())
        return (None if da is None else da.view(sa)), (None if db is None else db.view(sb)), None


class LossTermsFn(Function):
    """sum_t w_t * x_t[mask_t].mean() of a trainer's loss assembly (moda.py:540-705) as one launch each way (moda_loss_terms).
    spec: one (weight, mask, kind) per value tensor -- mask None (every row), or an (n,) / (n, 1) tensor: float selected where
    > 0 (kind '>0'), bool (kind 'bool').  -> (total 0-dim, terms (T,) detached weighted terms)."""

    @staticmethod
    def forward(ctx, spec, *xs):
        T = len(xs)
        dev = xs[0].device
        out = torch.empty((1 + 2 * T,), device=dev)
        arr = (L.LossTerm * T)()
        keep = []
        for t, (x, (w, mask, kind)) in enumerate(zip(xs, spec)):
            x2 = _f32(x)
            n = x2.shape[0] if x2.dim() > 0 else 1
            x2 = x2.reshape(n, -1)
            mk, mp = 0, None
            if mask is not None:
                if kind == "bool":
                    m = mask.reshape(-1).contiguous()
                    if m.dtype != torch.bool:
                        raise TypeError("kind 'bool' needs a bool mask")
                    mk = 2
                else:
                    m = _f32(mask).reshape(-1)
                    mk = 1
                if m.numel() != n:
                    raise ValueError(f"term {t}: {n} rows but a mask of {m.numel()}")
                mp = L.ptr(m)
                keep.append(m)
            keep.append(x2)
            arr[t] = L.LossTerm(x=L.ptr(x2), mask=mp, dx=None, n=n, k=x2.shape[1], mask_kind=mk, weight=float(w), reserved=0)
        L.call("moda_loss_terms", arr, T, L.ptr(out), None, L.stream())
        ctx.arr, ctx.keep, ctx.shapes, ctx.T = arr, keep, [tuple(x.shape) for x in xs], T
        ctx.save_for_backward(out)
        total, terms = out[0], out[1:1 + T]
        ctx.mark_non_differentiable(terms)
        return total, terms

    @staticmethod
    def backward(ctx, g, _g_terms):
        out, = ctx.saved_tensors
        dxs = []
        for t in range(ctx.T):
            need = ctx.needs_input_grad[1 + t]
            dx = torch.empty((ctx.arr[t].n, ctx.arr[t].k), device=out.device) if need else None
            ctx.arr[t].dx = L.ptr(dx)
            dxs.append(dx)
        L.call("moda_loss_terms", ctx.arr, ctx.T, L.ptr(out), L.ptr(_f32(g).reshape(1)), L.stream())
        return (None,) + tuple(None if d is None else d.view(sh) for d, sh in zip(dxs, ctx.shapes))


class S3imFn(Function):
    """S3IM.forward on already gathered index tables (loss_utils.py:575-702): 1 - mean SSIM of the (3, H, Wt) virtual patch.
    rgb (N,3) rendered colours, tar (N,3) observed colours, mask (N,1); index (H*Wt,) int32."""

    @staticmethod
    def forward(ctx, rgb, tar, mask, index, patch_h):
        r, t, m = _f32(rgb), _f32(tar), _f32(mask).reshape(-1)
        idx = L.dev(index, torch.int32)
        out = torch.empty((1,), device=r.device)
        L.call("moda_s3im", L.ptr(r), L.ptr(t), L.ptr(m), r.shape[0], L.ptr(idx), int(patch_h), idx.numel() // int(patch_h),
               L.ptr(out), None, None, L.stream())
        ctx.save_for_backward(r, t, m, idx)
        ctx.patch_h = int(patch_h)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        r, t, m, idx = ctx.saved_tensors
        d = zeros_like(r)
        L.call("moda_s3im", L.ptr(r), L.ptr(t), L.ptr(m), r.shape[0], L.ptr(idx), ctx.patch_h, idx.numel() // ctx.patch_h, None,
               L.ptr(_f32(g).reshape(1)), L.ptr(d), L.stream())
        return d, None, None, None, None


class LogSigLossFn(Function):
    """scale * sum_i -logsigmoid(sign * x_i) * (w_i | 1)  (visibility_loss, loss_utils.py:140,145) -> 0-dim tensor."""

    @staticmethod
    def forward(ctx, x, w, sign, scale):
        x1 = _f32(x).reshape(-1)
        w1 = None if w is None else _f32(w).reshape(-1)
        out = zeros((1,), x1.device)
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


class SplitRowsFn(Function):
    """x (M, ...) -> (x[:n], x[n:]) as views.  What it saves is autograd's backward of two slices of one tensor: two full-size
    zero fills, two copies and an add (16 MB each for the batched feature-net evaluation of a training step); here the two
    gradients are concatenated once."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(MATERIALIZE_GRADS)   # off: an output nobody differentiates arrives as None, not as a zero fill
        ctx.n = int(n)
        ctx.shape = tuple(x.shape)
        return x[:ctx.n], x[ctx.n:]

    @staticmethod
    def backward(ctx, ga, gb):
        n, shape = ctx.n, ctx.shape
        if ga is None and gb is None:
            return None, None
        ref = ga if ga is not None else gb
        if ga is None:
            ga = torch.zeros((n,) + shape[1:], device=ref.device, dtype=ref.dtype)
        if gb is None:
            gb = torch.zeros((shape[0] - n,) + shape[1:], device=ref.device, dtype=ref.dtype)
        return torch.cat([ga, gb], 0), None


class VisPairLossFn(Function):
    """visibility_loss's two terms on ONE logit vector (loss_utils.py:139-146): x = [negatives (n_neg) | positives],
    0.1 / n * sum -logsigmoid(-x_neg) + 1 / n * sum -logsigmoid(x_pos) w_pos with n = n_neg -> 0-dim.  Two launches each way on
    the halves of x and of dx; no slice nodes in the graph."""

    @staticmethod
    def forward(ctx, x, w_pos, n_neg):
        x1 = _f32(x).reshape(-1)
        w1 = _f32(w_pos).reshape(-1)
        n = int(n_neg)
        if x1.numel() - n != w1.numel():
            raise ValueError("VisPairLossFn: one weight per positive")
        out = zeros((1,), x1.device)
        L.call("moda_logsig_loss", L.ptr(x1), None, n, -1.0, 0.1 / n, L.ptr(out), None, None, L.stream())
        L.call("moda_logsig_loss", L.ptr(x1[n:]), L.ptr(w1), w1.numel(), 1.0, 1.0 / n, L.ptr(out), None, None, L.stream())
        ctx.save_for_backward(x1, w1)
        ctx.meta = (tuple(x.shape), n)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        x1, w1 = ctx.saved_tensors
        shape, n = ctx.meta
        dx = torch.empty_like(x1)
        gp = L.ptr(_f32(g).reshape(1))
        L.call("moda_logsig_loss", L.ptr(x1), None, n, -1.0, 0.1 / n, None, gp, L.ptr(dx), L.stream())
        L.call("moda_logsig_loss", L.ptr(x1[n:]), L.ptr(w1), w1.numel(), 1.0, 1.0 / n, None, gp, L.ptr(dx[n:]), L.stream())
        return dx.view(shape), None, None


class GradBucket:
    """One flat fp32 gradient buffer for a set of parameters, `p.grad` being views of it (what DistributedDataParallel's
    `gradient_as_bucket_view=True` does with its buckets).  Opt-in.  While a parameter's `.grad` is its view of a bucket,
    NerfFn's backward ADDS that network's weight gradients straight into the bucket (its GEMMs accumulate anyway) instead of
    handing autograd one fresh tensor per parameter per call: a network evaluated three times in a step costs no
    per-parameter `add` launches and no per-call zero fill, `zero()` is one memset, and `all_reduce()` exchanges the
    bucket as it lies (xGMI rings are per-link bound: one large message).  Gradients that reach a parameter by any other
    route (autograd's own accumulation) land in the same views.  Note: autograd hooks on these parameters do not fire for
    the directly written part, and `torch.autograd.grad(out, [x])` through a bound network still adds that network's weight
    gradients into the views (a custom Function cannot see which of its differentiable inputs a backward call asked for): wrap such
    probes in `with bucket.detached():`.  A parameter frozen later (`requires_grad_(False)`) switches its
    network back to returned gradients, so it receives nothing."""

    def __init__(self, params, extra=0):
        """extra: floats appended behind the gradients (`self.extra`, a view) that travel in the same collective -- the trainer's
        loss sums: a step then issues ONE all-reduce."""
        self.params = []
        seen = set()
        for p in params:
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                self.params.append(p)
        if not self.params:
            raise ValueError("GradBucket: no trainable parameter")
        dev = self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("GradBucket: parameters must be fp32 tensors on one device")
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4                    # 16-byte aligned views
        self.n_grad = off
        self.flat = zeros((off + (int(extra) + 3) // 4 * 4,), dev)
        self.extra = self.flat[off:off + int(extra)]
        self.attach()

    def attach(self):
        """(Re)install the views, e.g. after `optimizer.zero_grad(set_to_none=True)`."""
        for p, off in zip(self.params, self.offsets):
            v = self.flat[off:off + p.numel()].view(p.shape)
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v
            p._moda_bucket_ptr = v.data_ptr()

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, dist=None, world=1, average=True, force=False):
        """DDP semantics: the mean over ranks, in place, one collective (the `extra` floats included: sums that are read as a
        ratio are unaffected by the division).  average=False leaves the division to the caller (`scale`), e.g. inside a
        captured graph that follows the collective."""
        if world > 1 or force:             # force: a one-rank process group (sharding.COLLECTIVES_AT_WORLD_1)
            dist.all_reduce(self.flat)
            if average and world > 1:
                self.flat /= world
        return self.flat.numel()

    def scale(self, world):
        if world > 1:
            self.flat.mul_(1.0 / world)

    def detached(self):
        """`with bucket.detached():` -- inside, backward passes through the bound networks RETURN their parameter gradients to
        autograd instead of adding them into the bucket: what a `torch.autograd.grad(out, [xyz])` probe between `zero()` and the
        optimiser step needs (autograd then drops the gradients it was not asked for; written straight into the views they would
        pollute what is all-reduced next).  The views themselves stay installed as `.grad`."""
        return _BucketDetached(self)

    def check_same_layout_on_all_ranks(self, dist, world):
        """Every rank must exchange the same flat layout: all-reduce the element count with MIN and MAX once and compare (a rank
        whose first step reached a different parameter set would otherwise hang or corrupt the gradient all-reduce)."""
        if world <= 1:
            return
        n = torch.tensor([float(self.flat.numel()), -float(self.flat.numel())], device=self.flat.device, dtype=torch.float64)
        dist.all_reduce(n, op=dist.ReduceOp.MIN)
        lo, hi = int(n[0].item()), -int(n[1].item())
        if lo != hi:
            raise RuntimeError(f"GradBucket: ranks disagree on the bucket's size ({lo} .. {hi} floats) -- the parameter sets that "
                               "received a gradient in the first step differ between ranks")


class _BucketDetached:
    def __init__(self, bucket):
        self.bucket = bucket

    def __enter__(self):
        self.saved = [getattr(p, "_moda_bucket_ptr", None) for p in self.bucket.params]
        for p in self.bucket.params:
            p._moda_bucket_ptr = None
        return self.bucket

    def __exit__(self, *exc):
        for p, v in zip(self.bucket.params, self.saved):
            p._moda_bucket_ptr = v
        return False


def _bucket_grads(params, unused=()):
    """The `.grad` views of `params` when every one of them is bound to a GradBucket (else None).  Indices in `unused` are the
    heads this network does not evaluate (nerf.py:179-180, 190-197): they receive no gradient and need no view (None)."""
    out = []
    for i, p in enumerate(params):
        g = p.grad
        if g is None and i in unused:
            out.append(None)
            continue
        if g is None or getattr(p, "_moda_bucket_ptr", None) != g.data_ptr() or g.dtype != torch.float32 or not g.is_contiguous():
            return None
        out.append(g)
    return out


# ---- whole-network training Function ---------------------------------------------------------------------------------
class NerfSpec:
    """Static description of one NeRF module for NerfFn (nerf.py:84-140)."""

    def __init__(self, D, W, P, C1, Cd, n_out, raw_feat, n_freq, window, sigma_only=False):
        self.D, self.W, self.P, self.C1, self.Cd, self.n_out = D, W, P, C1, Cd, n_out
        self.raw_feat, self.n_freq, self.window, self.sigma_only = raw_feat, n_freq, list(window), sigma_only
        self.Pp = (P + 3) // 4 * 4          # PE columns padded so that every GEMM operand row is 16-byte aligned


class NerfFn(Function):
    """Embedding + NeRF.forward (nerf.py:35-75, 147-198) for the training route as ONE autograd node.

    Inputs: xyz (M,3) sample positions; code (R1,C1)|None the per-ray part of `input_xyz` (pose code); dir_src
    (Rd,Cd)|None the per-ray `input_dir` (direction embedding ++ env / appearance codes); then the module's
    parameters.  Per-ray inputs are never expanded to samples: they go through their weight columns once per ray and
    enter the layer as a row-group bias (their gradients are per-ray segment sums).  The skip layer reads
    cat[PE, h] from its two sources in place.  Forward: one pipelined fp32-MFMA GEMM per layer with the bias / ReLU /
    sigmoid epilogue.  Backward: per layer dW (split-K), db, and dX with the ReLU mask of the layer below fused into
    its epilogue.  Activations are kept in fp32 (exact-parity training, as the reference's autograd).
    The launch schedule lives in the library (moda_nerf_train_fwd / _bwd): one host call each way per network."""

    @staticmethod
    def _desc(sp, M, R1, Rd, flags=None):
        d = L.NerfTrainDesc(D=sp.D, W=sp.W, P=sp.P, C1=sp.C1, Cd=sp.Cd, n_out=sp.n_out, raw_feat=int(sp.raw_feat),
                            sigma_only=int(sp.sigma_only), n_freq=sp.n_freq,
                            reserved=_gemm_flags() if flags is None else flags, M=M, R1=R1, Rd=Rd)
        for k in range(16):
            d.window[k] = sp.window[k] if k < sp.n_freq else 0.0
        return d

    @staticmethod
    def forward(ctx, spec, xyz, code, dir_src, *params):
        sp = spec
        x = _f32(xyz).reshape(-1, 3)
        M = x.shape[0]
        dev = x.device
        pr = [_f32(p) for p in params]
        cd = None if code is None else _f32(code).reshape(-1, sp.C1)
        ds = None if (dir_src is None or sp.sigma_only) else _f32(dir_src).reshape(-1, sp.Cd)
        R1 = 1 if cd is None else cd.shape[0]
        Rd = 1 if ds is None else ds.shape[0]
        pack = getattr(sp, "pack", None)
        # (the same shape limits as moda_nerf_train_fwd_fused / moda_mlp_dump_fwd: anything else takes the per-layer forward)
        fused = (pack is not None and _TRAIN_PRECISION == "bf16" and FUSED_TRAIN_FORWARD and not sp.sigma_only
                 and sp.W in (64, 128, 256) and 5 <= sp.D <= 8 and sp.n_freq <= 10 and 1 <= sp.n_out <= 64
                 and M * sp.W * 4 < 2 ** 32)         # the dump kernel addresses a layer with 32-bit byte offsets
        flags = _gemm_flags() | (_STORE_FLAG if fused and TRAIN_BF16_STORE else 0)
        d = NerfFn._desc(sp, M, R1, Rd, flags)
        lib = L.load()
        nws = lib.moda_nerf_train_ws_floats(L._c.byref(d))
        if nws < 0:
            raise NotImplementedError(f"NerfFn: unsupported network shape D={sp.D} W={sp.W}")
        ws = torch.empty((nws,), device=dev, dtype=torch.float32)
        n_cols = 1 if sp.sigma_only else sp.n_out + (0 if sp.raw_feat else 1)
        out = torch.empty((M, n_cols), device=dev, dtype=torch.float32)
        pp = (L._P * len(pr))(*[p.data_ptr() for p in pr])
        if fused:
            # throughput mode: ONE launch of the fused bf16 PE+MLP kernel writes every layer's activations into ws
            stream, bias, bd_folded = pack
            L.call("moda_nerf_train_fwd_fused", L._c.byref(d), L.ptr(x), L.ptr(cd), L.ptr(ds), pp, L.ptr(stream), L.ptr(bias),
                   L.ptr(bd_folded), L.ptr(ws), L.ptr(out), L.stream())
        else:
            L.call("moda_nerf_train_fwd", L._c.byref(d), L.ptr(x), L.ptr(cd), L.ptr(ds), pp, L.ptr(ws), L.ptr(out), L.stream())
        ctx.spec, ctx.M, ctx.flags = sp, M, flags
        ctx.save_for_backward(x, cd, ds, ws, out, *pr)
        return out

    @staticmethod
    def backward(ctx, g_out):
        sp, M = ctx.spec, ctx.M
        sv = ctx.saved_tensors
        x, cd, ds, ws, out = sv[:5]
        pr = list(sv[5:])
        dev = x.device
        R1 = 1 if cd is None else cd.shape[0]
        Rd = 1 if ds is None else ds.shape[0]
        d = NerfFn._desc(sp, M, R1, Rd, ctx.flags)        # the precision / storage mode the forward ran in
        lib = L.load()
        g = _f32(g_out)
        scratch = torch.empty((lib.moda_nerf_train_scratch_floats(L._c.byref(d)),), device=dev, dtype=torch.float32)
        # Parameter gradients: every write of the library ADDS.  Bound to a GradBucket (opt-in), they go straight into the
        # parameters' `.grad` views and autograd gets nothing to accumulate; otherwise into one zero-filled buffer whose views
        # are returned.
        D = sp.D
        unused = tuple(range(2 * D + 2, 2 * D + 8)) if sp.sigma_only else ((2 * D, 2 * D + 1) if sp.raw_feat else ())
        # The direct route only while every parameter of the network still wants a gradient: a parameter frozen after the bucket
        # was built (requires_grad_(False): it keeps its view) makes the whole network fall back to returned gradients, which
        # autograd hands only to the parameters that still want them.  (`needs_input_grad` is fixed at FORWARD time from the
        # inputs' requires_grad, not per backward call: `torch.autograd.grad(out, [xyz])` through a bucket-bound network therefore
        # still adds the weight gradients into `.grad` -- the caveat GradBucket's docstring documents; `with bucket.detached():` there.)
        n_fixed = 4                                   # spec, xyz, code, dir_src precede the parameters in forward()'s arguments
        wanted = all(ctx.needs_input_grad[n_fixed + i] for i in range(len(pr)) if i not in unused)
        objs = getattr(sp, "param_objs", None)
        direct = _bucket_grads(objs, unused) if (objs and wanted) else None
        sizes = [p.numel() for p in pr]
        n_code = 0 if cd is None else cd.numel()
        if direct is not None:
            dummy = torch.empty((max(sizes),), device=dev, dtype=torch.float32) if any(g is None for g in direct) else None
            grads = [dummy if g is None else g for g in direct]       # (never written: the heads are not evaluated)
            d_code = None if cd is None else zeros_like(cd)
        else:
            flat = zeros((sum(sizes) + n_code,), dev)
            grads, off = [], 0
            for p, n in zip(pr, sizes):
                grads.append(flat[off:off + n].view(p.shape))
                off += n
            d_code = None if cd is None else flat[off:off + n_code].view(cd.shape)
        d_dir = None if ds is None else torch.empty_like(ds)
        d_xyz = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        pp = (L._P * len(pr))(*[p.data_ptr() for p in pr])
        gp = (L._P * len(pr))(*[t.data_ptr() for t in grads])
        L.call("moda_nerf_train_bwd", L._c.byref(d), L.ptr(x), L.ptr(cd), L.ptr(ds), pp, L.ptr(ws), L.ptr(out), L.ptr(g),
               L.ptr(scratch), gp, L.ptr(d_xyz), L.ptr(d_code), L.ptr(d_dir), L.stream())
        if direct is not None:
            return (None, d_xyz, d_code, d_dir) + (None,) * len(pr)
        if sp.sigma_only:                       # heads that were not evaluated get no gradient
            for i in range(2 * D + 2, 2 * D + 8):
                grads[i] = None
        elif sp.raw_feat:
            grads[2 * D] = grads[2 * D + 1] = None
        return (None, d_xyz, d_code, d_dir) + tuple(grads)


class FanOutFn(Function):
    """x -> n aliases of x, each to be consumed by ONE node.  Autograd sums the gradients of a tensor that feeds k nodes with k - 1
    `add` launches (one out-of-place, k - 2 in place); here the k gradients arrive together and are summed by one launch
    (`moda_sum_tensors`, argument order: deterministic).  Aliases nobody differentiates cost nothing (their gradient is None)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view(x.shape) for _ in range(int(n)))

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        live = [_f32(g) for g in live]
        if any(g.shape != live[0].shape or g.data_ptr() % 16 for g in live):
            out = live[0]
            for g in live[1:]:
                out = out + g
            return out, None
        out = torch.empty_like(live[0])
        done = 0
        while done < len(live):                      # eight operands per launch (never more than one launch in MoDA's step)
            part = ([out] if done else []) + live[done:done + (7 if done else 8)]
            done += 7 if done else 8
            L.call("moda_sum_tensors", (L._P * len(part))(*[g.data_ptr() for g in part]), len(part), out.numel(), L.ptr(out), L.stream())
        return out, None


class Fan:
    """`take = Fan(x, n)`: every `take()` hands out the next alias of x (see FanOutFn); x itself when no gradient can flow."""

    def __init__(self, x, n):
        self.x, self.i = x, 0
        self.out = FanOutFn.apply(x, n) if (torch.is_tensor(x) and torch.is_grad_enabled() and x.requires_grad) else None

    def __call__(self):
        if self.out is None:
            return self.x
        if self.i >= len(self.out):
            raise RuntimeError("Fan: more consumers than aliases")
        v = self.out[self.i]
        self.i += 1
        return v


class fan_scope:
    """`with fan_scope():` -- inside, `fanned(t)` hands out a fresh alias of `t` per call, all aliases of one tensor coming from ONE
    FanOutFn node, so that however many nodes of the training route consume `t` (the warped sample positions feed eight), its
    gradient is formed by one launch.  Outside a scope, or for tensors no gradient flows to, `fanned(t)` is `t`."""
    active = None
    WIDTH = 8                      # aliases per tensor; a ninth consumer simply gets the tensor itself (autograd adds that one)

    def __enter__(self):
        self.saved, fan_scope.active = fan_scope.active, {}
        return self

    def __exit__(self, *exc):
        fan_scope.active = self.saved
        return False


def fanned(t, key=None):
    reg = fan_scope.active
    if reg is None or not torch.is_tensor(t) or not (torch.is_grad_enabled() and t.requires_grad):
        return t
    k = id(t if key is None else key)
    ent = reg.get(k)
    if ent is None:
        ent = reg[k] = (t if key is None else key, Fan(t, fan_scope.WIDTH))       # (the tensor is held: its id stays its own)
    fan = ent[1]
    return fan() if fan.i < fan_scope.WIDTH else t


class ExpandRowsFn(Function):
    """(F, C) per-frame rows -> (F*k, C): every row repeated for the k consecutive rays of its frame (what moda.update_rays
    does with .repeat, moda.py:1281-1311).  Forward is a copy; backward sums each frame's k gradient rows (moda_segsum_f32)."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k = k
        x2 = _f32(x)
        return x2[:, None, :].expand(x2.shape[0], k, x2.shape[1]).reshape(-1, x2.shape[1])

    @staticmethod
    def backward(ctx, g):
        g2 = _f32(g)
        C = g2.shape[1]
        out = torch.empty((g2.shape[0] // ctx.k, C), device=g2.device, dtype=torch.float32)
        L.call("moda_segsum_f32", L.ptr(g2), out.shape[0], ctx.k, C, C, L.ptr(out), C, L.stream())
        return out, None
