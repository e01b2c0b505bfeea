"""`render_rays` and helpers with the reference's call surface (nnutils/rendering.py:19-623), on the HIP library.

Scope: ray sampling, the bones / neudbs warp, the MLP stack, SDF->density compositing,
hierarchical resampling, and every result-dict key `inference_deform` produces in MoDA's configuration, including
the per-ray heads behind compositing (paired-frame flow rendering, CSE feature matching, keypoint reprojection,
visibility loss, uncertainty head, img / sil / flo / feature loss terms; rendering.py:410-578, moda_amd/loss_utils.py).
Branches MoDA's recipe never takes (lbs, flowbw/flowfw) raise NotImplementedError instead of silently skipping; the
default-off s3im_loss term (rendering.py:528-532, loss_utils.py:648-702) is built (loss_utils.s3im_loss).

Random tensors: the reference draws torch.rand / rand_like / randn internally (rendering.py:82,193,389,607).
They are drawn here on the rays' device in the same order and shapes; `rng` (dict) can inject any of
'perturb_rand' (N,S), 'pdf_u' (N,S/2), 'symm_rand' / 'symm_rand_pre' (N,S,1) uniforms, 'noise_raw' /
'noise_raw_pre' (N,S) standard normals, 'feat_noise' (1,8000,3) / 'vis_neg_rand' (1,N*S,3) of the loss heads
(loss_utils.py:306, :137), for bit-reproducible comparisons against the CPU oracle.
"""
import os

import torch

from . import _lib as L
from .geom_utils import bone_transform, warp
from .nerf import get_precision, hot_precision, precision_scope

# Throughput mode: run skin MLP -> softmax -> DQS as one kernel per warp (NeRF.fused_warp).  False keeps the round-1
# two-kernel route (MLP writes the (N,B,S) logits, the warp kernel reads them): used for A/B timing and by the tests that
# compare the two routes.
FUSED_WARP = os.environ.get("MODA_FUSED_WARP", "1") != "0"
# fp16 mode: what the two pieces outside the hot loop run in (see nerf.default_precision): the 128-wide feature network
# (`nerf_feat`, raw outputs composited into the rendered features) and the hierarchical pre-pass (its weights go through the
# inverse CDF of sample_pdf).  'bf16x3' is the measured-safe default of round 4; tools/fp16_cfg5_probe.py measures the others.
FP16_FEAT_PRECISION = os.environ.get("MODA_FP16_FEAT", "bf16x3")
FP16_PREPASS_PRECISION = os.environ.get("MODA_FP16_PREPASS", "bf16x3")
FP16_PREPASS_WARP = os.environ.get("MODA_FP16_PREPASS_WARP", "")      # with an fp16 pre-pass: the precision of ITS skin + warp kernel ("" = fp16)
# fp16 mode, split-bf16 pre-pass: what ITS skin + warp runs in.  "" = the two-kernel split-bf16 route (skin MLP writing the
# (N, B, S) logits + the warp kernel: 1.9 ms at config 5); "fp16" = the one-kernel fp16 skin + warp the final pass uses (0.7 ms)
FP16_X3_PREPASS_WARP = os.environ.get("MODA_FP16_X3_PREPASS_WARP", "fp16")     # (measured: tools/fp16_cfg5_probe.py, per-element 0.61 either way)
# fp16 mode: the feature network when NO head of the call consumes the rendered features (no rays['feats_at_samp']: the reference
# then composites them and drops them, rendering.py:395 -> :573-578; here they come back as result['feat_rnd'], not a reference key)
FP16_FEAT_UNCONSUMED_PRECISION = os.environ.get("MODA_FP16_FEAT_UNCONSUMED", "fp16")
_PREPASS_OF_FP16 = False               # set by render_rays around the hierarchical pre-pass of an fp16-mode call
# Hierarchical sampling, inference route: the merged final pass (rendering.py:116) evaluates every network at S/2 coarse + S/2
# importance depths -- and the coarse half is exactly where the no-grad pre-pass (:96-104) has just evaluated the same pointwise
# functions (backward warp, 8 x 256 network).  With this on, the pre-pass keeps what it computed (warped positions, colour +
# density: it evaluates the colour branch too, +9 % of its MACs), the final pass runs warp + network on the importance depths only,
# and `moda_merge_rows` puts the two halves in depth order (`moda_merge_index`: the sorted depths with their origin).  Same
# results (fp32 / bf16 / bf16x3: the same kernels on the same points; fp16 mode: the coarse half comes from the split-bf16
# pre-pass, i.e. closer to fp32), a quarter of the call's 8 x 256 evaluations gone.  Not with symm_shape (the two passes draw
# different flips, :389), nerf_dis, or early termination.  MODA_REUSE_COARSE=0: every depth evaluated in the final pass (A/B).
REUSE_COARSE = os.environ.get("MODA_REUSE_COARSE", "1") != "0"
# (Rounds 3-4 carried an opt-in switch that ran the feature-matching head on a side stream, MODA_HEAD_STREAMS=1: 1 % of the
# captured step, and nerf_feat's gradients 1e-4 ... 5e-4 off in ~40 % of fresh processes -- tensors crossing the two streams went
# back to the allocator pool of the stream that made them while the other could still read them.  A switch that silently corrupts
# gradients does not ship: deleted in round 5.)
ROW_RUNS = os.environ.get("MODA_ROW_RUNS", "1") != "0"      # 0: per-frame work on every per-ray copy, as before round 4 (A/B)
# mode -> precision of the one-kernel skin + warp route (absent: two-kernel route)
WARP_PRECISION = {"bf16": "bf16", "fp16": os.environ.get("MODA_FP16_WARP", "fp16")}
if os.environ.get("MODA_X3_FUSED_WARP", "0") == "1":
    WARP_PRECISION["bf16x3"] = "bf16x3"
# Throughput mode: compositing as the epilogue of the 8 x 256 kernel (NeRF.fused_composite, moda_mlp_composite_fwd) when the call
# is its plain form (no nerf_feat / clip bound / visibility mask / rgb_filter / termination, 32-256 samples per ray).  The two
# routes are bit-identical (tests).  OFF by default: measured on one box, interleaved (profiles/r03/fused_composite_ab.md), the
# fused form takes the (N,S,4) round trip (0.54 GB) and a launch out of the step but is 0.8-1.2 % SLOWER -- the epilogue's
# transcendental chain, scans and two barriers run with the matrix pipe idle on all eight waves at once (the weight ring keeps
# them in step), where the separate kernel's 0.12 ms overlap nothing either but cost less.  MODA_FUSED_COMPOSITE=1 switches it on.
FUSED_COMPOSITE = os.environ.get("MODA_FUSED_COMPOSITE", "0") == "1"


# Frame-grouped ray layout: with rays['rays_per_frame'] = k the rays of one frame are consecutive
# and these keys may hold ONE row per frame, (N/k, C), instead of the per-ray repeats moda.update_rays builds
# (moda.py:1281-1311).  bone_rts and time_embedded are consumed per frame by the kernels (bone_transform on N/k rows, the
# warp's transform tables, the skin MLP's folded code rows); the others are expanded by a row copy.
FRAME_KEYS = ('bone_rts', 'bone_rts_target', 'bone_rts_dentrg', 'time_embedded', 'env_code', 'appearance_code', 'rtk_vec',
              'rtk_vec_target', 'rtk_vec_dentrg', 'vid_code')
_FRAME_NATIVE = ('bone_rts', 'time_embedded')


def _frame_layout(rays, N_rays, train):
    """-> (rays with per-frame rows expanded where the consumer is per-ray, k)."""
    k = rays.get('rays_per_frame', None)
    if k is None or int(k) <= 1:
        return rays, 1
    k = int(k)
    if N_rays % k:
        raise ValueError(f"rays_per_frame={k} does not divide {N_rays} rays")
    F = N_rays // k
    out = dict(rays)
    for key in FRAME_KEYS:
        t = rays.get(key, None)
        if not torch.is_tensor(t):
            continue
        t2 = t.reshape(-1, t.shape[-1])
        if t2.shape[0] == N_rays and F != N_rays:
            continue                                   # already per ray
        if t2.shape[0] != F:
            raise ValueError(f"rays['{key}']: expected {F} (per frame) or {N_rays} (per ray) rows, got {t2.shape[0]}")
        if key in _FRAME_NATIVE and not train:
            out[key] = t2
        elif torch.is_grad_enabled() and t2.requires_grad:
            from .autograd import ExpandRowsFn
            out[key] = ExpandRowsFn.apply(t2, k)
        else:
            out[key] = L.dev(t2)[:, None, :].expand(F, k, t2.shape[1]).reshape(N_rays, -1)
    return out, k


def _draw(rng, key, kind, shape, device):
    if rng is not None and key in rng and rng[key] is not None:
        return L.dev(rng[key]).reshape(shape)
    return (torch.rand if kind == "rand" else torch.randn)(shape, device=device)


_SKIP_DELTA = {}        # (device index, numel) -> Philox offset a torch.randn of that size consumes


def _skip_randn(shape, device):
    """Leave the default CUDA generator exactly where `torch.randn(shape, device=device)` would, without the launch.  The
    reference draws its density noise unconditionally (rendering.py:193) and multiplies it by noise_std; with noise_std == 0 the
    values are never read, but every later draw of the call (and of the caller) must see the generator state the reference's
    would see.  The generator is Philox: its state is (seed, offset), so the draw's effect is an offset increment that depends on
    the size alone -- measured once per size by a real draw, added from then on (16.8 M floats = 67 MB and 26 us per call at
    config 2).  Real draws under stream capture, where offsets are the graph's business."""
    n = 1
    for d in shape:
        n *= int(d)
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    gen = torch.cuda.default_generators[idx]
    key = (idx, n)
    delta = _SKIP_DELTA.get(key)
    if delta is None or torch.cuda.is_current_stream_capturing():
        off = gen.get_offset()
        torch.randn(shape, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _SKIP_DELTA[key] = gen.get_offset() - off
        return
    gen.set_offset(gen.get_offset() + delta)


def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5, u=None):
    """rendering.py:582-623: bins (N, S_+1), weights (N, S_) -> (N, N_importance) samples."""
    if eps != 1e-5:
        raise NotImplementedError("the kernel fixes eps=1e-5 (the only value the reference uses)")
    L.no_grad_only(bins, weights)
    b = L.dev(bins)
    w = L.dev(weights)
    n, nb = b.shape
    assert w.shape == (n, nb - 1)
    if not det and u is None:
        u = torch.rand(n, N_importance, device=b.device)            # :607
    uu = None if det else L.dev(u)
    out = torch.empty((n, N_importance), device=b.device, dtype=torch.float32)
    L.call("moda_sample_pdf_fwd", L.ptr(b), L.ptr(w), L.ptr(uu), n, nb, N_importance, L.ptr(out), L.stream())
    return out


def _merge_sorted(a, b):
    n, la = a.shape
    lb = b.shape[1]
    out = torch.empty((n, la + lb), device=a.device, dtype=torch.float32)
    L.call("moda_merge_sort_fwd", L.ptr(a), la, L.ptr(b), lb, n, L.ptr(out), L.stream())
    return out


def _merge_index(za, zb):
    """-> (z (N, La+Lb) = sort(cat(za, zb)), src (N, La+Lb) int32: index into the concatenation each sorted depth came from)."""
    n, la = za.shape
    lb = zb.shape[1]
    z = torch.empty((n, la + lb), device=za.device, dtype=torch.float32)
    src = torch.empty((n, la + lb), device=za.device, dtype=torch.int32)
    L.call("moda_merge_index_fwd", L.ptr(L.dev(za)), la, L.ptr(L.dev(zb)), lb, n, L.ptr(z), L.ptr(src), L.stream())
    return z, src


def _merge_rows(src, a, b):
    """a (N, La, C), b (N, Lb, C) -> (N, La+Lb, C): row p of ray n is a[n, src] if src < La else b[n, src - La]."""
    n, l = src.shape
    la, c = a.shape[1], a.shape[2]
    out = torch.empty((n, l, c), device=a.device, dtype=torch.float32)
    L.call("moda_merge_rows_fwd", L.ptr(src), n, l, la, c, L.ptr(L.dev(a)), L.ptr(L.dev(b)), L.ptr(out), L.stream())
    return out


def joint_row_runs(*tensors):
    """run_start (R,) int32 of the runs of consecutive rows that are bit-identical in EVERY given (R, c_i) tensor
    (`moda_row_runs_multi`): the reference's ray layout repeats each frame's bone_rts / time_embedded / env_code row for all of
    the frame's rays (moda.py:1302-1310); per-frame work -- bone_transform, the skin net's code folds, the warp kernels' operand
    tables -- is then done at a run's first row only.  No host synchronisation: the partition stays on the device."""
    ts = [L.dev(t).reshape(t.shape[0], -1) for t in tensors]
    R = ts[0].shape[0]
    if any(t.shape[0] != R for t in ts) or not 1 <= len(ts) <= 4:
        raise ValueError("joint_row_runs: one to four tensors with the same number of rows")
    runs = torch.empty((R,), device=ts[0].device, dtype=torch.int32)
    ws = torch.empty(((R + 255) // 256,), device=ts[0].device, dtype=torch.int32)
    L.call("moda_row_runs_multi", len(ts), (L._P * len(ts))(*[t.data_ptr() for t in ts]), (L._I64 * len(ts))(*[t.shape[1] for t in ts]),
           R, L.ptr(runs), L.ptr(ws), L.stream())
    return runs


def composite(rgbsigma, feat, z_vals, rays_d, beta, noise=None, xyz=None, clip_bound=None, vis_pred=None, cyc=None,
              rgb_filter_scale=0.0, n_live=None, term_tau=0.0, want_visibility=True, want_weights=True):
    """inference() tail (rendering.py:183-237) -> dict(rgb, feat, depth, sil, weights, visibility, vis_out, cyc_out).
    n_live / term_tau: opt-in early ray termination (moda_composite_fwd); then also 'n_used' (N,) int32."""
    N, S = z_vals.shape
    dev_ = z_vals.device
    F = 0 if feat is None else feat.shape[-1]
    o = {
        "rgb": torch.empty((N, 3), device=dev_), "depth": torch.empty((N,), device=dev_),
        "sil": torch.empty((N,), device=dev_), "weights": torch.empty((N, S), device=dev_) if want_weights else None,
        "visibility": torch.empty((N, S), device=dev_) if want_visibility else None,
        "feat": torch.empty((N, F), device=dev_) if F else None,
        "vis_out": torch.empty((N,), device=dev_) if vis_pred is not None else None,
        "cyc_out": torch.empty((N,), device=dev_) if cyc is not None else None,
    }
    cb = None
    if clip_bound is not None:
        cb = torch.as_tensor(clip_bound, dtype=torch.float32).reshape(3).to(dev_)   # :211
    early = n_live is not None or term_tau > 0
    o["n_used"] = torch.empty((N,), device=dev_, dtype=torch.int32) if early else None
    nl = None if n_live is None else L.dev(n_live, torch.int32).reshape(N)
    L.call("moda_composite_fwd", L.ptr(rgbsigma), L.ptr(feat), F, L.ptr(z_vals), L.ptr(rays_d), L.ptr(beta),
           L.ptr(noise), L.ptr(xyz), L.ptr(cb), L.ptr(vis_pred), L.ptr(cyc), float(rgb_filter_scale), N, S,
           L.ptr(o["rgb"]), L.ptr(o["feat"]), L.ptr(o["depth"]), L.ptr(o["sil"]), L.ptr(o["weights"]),
           L.ptr(o["visibility"]), L.ptr(o["vis_out"]), L.ptr(o["cyc_out"]), L.ptr(nl), float(term_tau), L.ptr(o["n_used"]),
           L.stream())
    return o


def inference(models, embedding_xyz, xyz_, dir_, dir_embedded, z_vals, N_rays, N_samples, chunk, noise_std,
              env_code=None, appearance_code=None, weights_only=False, clip_bound=None, vis_pred=None,
              scale_rgb=1.3, rgb_filter=False, flip=None, noise_raw=None, cyc=None, _full=False, n_live=None, term_tau=0.0,
              _want_visibility=True, _want_weights=True, _feat_consumed=True, _keep=False, _reuse=None):
    """rendering.py:124-237.  dir_embedded is per ray (N_rays, 27).  Returns the reference's 6-tuple
    (rgb, feat, depth, weights, visibility, sil) (or the composite dict with _full=True)."""
    nerf_sdf = models['coarse']
    xyz = L.dev(xyz_).reshape(N_rays, N_samples, 3)
    z = L.dev(z_vals)
    side = [L.dev(dir_embedded).reshape(N_rays, -1)]
    if env_code is not None:
        side.append(L.dev(env_code).reshape(N_rays, -1))
    if appearance_code is not None:
        side.append(L.dev(appearance_code).reshape(N_rays, -1))
    dir_src = torch.cat(side, -1)                                             # geom_utils.py:33-50 column order
    alpha = embedding_xyz.alpha
    nf = embedding_xyz.N_freqs
    if weights_only:
        # only the density is wanted (the weights of a hierarchical pre-pass): the colour branch is not evaluated
        # (nerf.py:179-180) and neither is the feature net -- the compositing weights do not depend on them
        sig = nerf_sdf.fused(xyz, n_freq=nf, alpha=alpha, flip=flip, sigma_only=True, precision=hot_precision())
        rgbsigma = torch.zeros((N_rays, N_samples, 4), device=xyz.device)
        rgbsigma[..., 3:] = sig
    else:
        if noise_raw is None:
            if noise_std == 0:
                _skip_randn((N_rays, N_samples), xyz.device)                                        # :193 (always drawn there)
                noise_raw = False                                                                   # (drawn, not materialised)
            else:
                noise_raw = torch.randn((N_rays, N_samples), device=xyz.device)                     # :193
        noise = None if noise_std == 0 else (L.dev(noise_raw).reshape(N_rays, N_samples) * noise_std)
        if (FUSED_COMPOSITE and get_precision() == "bf16" and 'nerf_feat' not in models.keys() and clip_bound is None
                and vis_pred is None and not rgb_filter and n_live is None and term_tau == 0 and appearance_code is None
                and not _keep and _reuse is None):
            o = nerf_sdf.fused_composite(xyz, z, L.dev(dir_), nerf_sdf.beta, n_freq=nf, alpha=alpha, dir_src=dir_src, flip=flip,
                                         noise=noise, cyc=cyc, want_visibility=_want_visibility or not _full)
            if o is not None:                                                                       # :159-237 in one kernel
                o["feat"] = torch.zeros_like(o["rgb"])                                              # :180
                if _full:
                    return o
                return o["rgb"], o["feat"], o["depth"], o["weights"], o["visibility"], o["sil"]
        live = n_live if (n_live is not None and N_samples % 32 == 0) else None     # whole 32-sample groups only
        if _reuse is not None:
            # REUSE_COARSE: the network on the importance depths alone; the coarse depths' rows are the pre-pass's
            rs_f = nerf_sdf.fused(_reuse['canon_fine'], n_freq=nf, alpha=alpha, dir_src=dir_src, precision=hot_precision())
            rgbsigma = _merge_rows(_reuse['src'], _reuse['rgbsigma'], rs_f)
        else:
            rgbsigma = nerf_sdf.fused(xyz, n_freq=nf, alpha=alpha, dir_src=dir_src, flip=flip, n_live=live,
                                      precision=hot_precision())                                       # :159
    feat = None
    if 'nerf_feat' in models.keys() and not weights_only and not _keep:
        # fp16 mode: raw network outputs are split-bf16 business (nerf.default_precision) -- where something consumes them.  With
        # no rays['feats_at_samp'] the rendered features reach no reference key (:573-578): the mode's own fp16 kernels then
        fp = None
        if get_precision() == "fp16":
            fp = FP16_FEAT_PRECISION if _feat_consumed else FP16_FEAT_UNCONSUMED_PRECISION
        feat = models['nerf_feat'].fused(xyz, n_freq=nf, alpha=alpha, flip=flip, precision=fp)   # :174-178
    if noise_raw is None:
        if noise_std == 0:
            _skip_randn((N_rays, N_samples), xyz.device)                                            # :193 (always drawn there)
        else:
            noise_raw = torch.randn((N_rays, N_samples), device=xyz.device)                         # :193
    noise = None if noise_std == 0 else (L.dev(noise_raw).reshape(N_rays, N_samples) * noise_std)
    o = composite(rgbsigma, feat, z, L.dev(dir_), L.dev(nerf_sdf.beta), noise=noise, xyz=xyz,
                  clip_bound=clip_bound, vis_pred=vis_pred, cyc=cyc,
                  rgb_filter_scale=float(scale_rgb) if rgb_filter else 0.0, n_live=n_live, term_tau=term_tau,
                  want_visibility=_want_visibility or not _full, want_weights=_want_weights or not _full)   # :171, 225-230
    if feat is None:
        o["feat"] = torch.zeros_like(o["rgb"])                                                      # :180
    if _keep:
        o["_rgbsigma"] = rgbsigma
    if _full:
        return o
    return o["rgb"], o["feat"], o["depth"], o["weights"], o["visibility"], o["sil"]


def _corresp_and_loss_heads(result, rays, models, opts, img_size, weights, xyz_canon, rgb, sil, embedding_xyz=None,
                            obj_bound=None, vis=None, feat_rnd=None, chunk=None, rng=None, dskin_rest=None, dskin_bns=False,
                            pts_tf=None, feat_grid=None):
    """Everything inference_deform computes behind compositing when fine_iter is set (rendering.py:410-437 feature
    matching + keypoint reprojection, 345-360 / 439-499 paired-frame correspondence and flow rendering, 475-477
    visibility loss, 501-516 uncertainty head, 518-578 per-ray loss terms), in the reference's order."""
    from . import autograd as A
    from . import loss_utils as LU
    F = A.fanned                           # (training route: one alias per consumer of a shared tensor, see _inference_deform_train)
    N_rays = xyz_canon.shape[0]            # (weights is None when no head below reads it: inference_deform(_want_weights=False))
    xys = L.dev(rays['xys']).reshape(N_rays, 2)
    has_bones = 'bones' in models.keys()
    is_training = models['coarse'].training
    pts_target = None
    if opts.use_corresp and 'rtk_vec_target' in rays.keys() and not opts.dist_corresp:
        pts_exp = LU.compute_pts_exp(F(weights), F(xyz_canon))                 # :411-415
        pts_target = LU.kp_reproj(pts_exp, models, embedding_xyz, rays, to_target=True, neudbs=opts.neudbs)
    feats_at = None
    if 'feats_at_samp' in rays.keys():                                         # :417-437
        feats_at = L.dev(rays['feats_at_samp'])
        pts_pred, pts_exp_f, feat_err, corr_err = LU.feat_match_loss(
            models['nerf_feat'], embedding_xyz, feats_at, F(xyz_canon), F(weights), obj_bound, opts.use_corr, opts.use_ot,
            is_training=is_training, rng=rng, grid=feat_grid)
        proj_err = LU.kp_reproj_loss(pts_pred, xys, models, embedding_xyz, rays, neudbs=opts.neudbs)
        result['pts_pred'], result['pts_exp'] = pts_pred, pts_exp_f
        result['feat_err'] = feat_err
        if opts.use_corr:
            result['corr_err'] = corr_err                                      # :434-435
        result['proj_err'] = proj_err / img_size * 2
        result['pts_exp_vis'], result['pts_pred_vis'] = pts_exp_f, pts_pred   # :467-469
    if is_training and 'nerf_vis' in models.keys():                            # :475-477
        result['vis_loss'] = LU.visibility_loss(models['nerf_vis'], embedding_xyz, xyz_canon, vis, obj_bound, chunk, rng=rng)
    flo_out = {}
    for tag, key in (("target", "flo"), ("dentrg", "fdp")):
        rk = 'rtk_vec_' + tag
        if rk not in rays.keys():
            continue
        if opts.dist_corresp:
            rtk = L.dev(rays[rk]).reshape(N_rays, 21)
            pts = xyz_canon                                                    # :253-254 clones of the samples
            if has_bones and ('bone_rts_' + tag) in rays.keys():
                pts = LU.forward_warp(F(xyz_canon), models, embedding_xyz, rays['bone_rts_' + tag], dskin=F(dskin_rest),
                                      dskin_bns=dskin_bns, pts_tf=pts_tf)      # :345-360 (nerf_dis: x* + dis(x*, rest))
            proj = A.ProjectFn.apply(pts, rtk)                                 # :439-461
            flo, valid = A.FlowRenderFn.apply(F(weights), proj, xys, img_size)    # :480-483, 491-494
        else:
            if pts_target is None or tag != "target":
                raise NotImplementedError("flow from a reprojected expected point needs opts.use_corresp and the target "
                                          "frame (rendering.py:411-415, 485; pts_dentrg is never defined there)")
            flo = (pts_target.reshape(N_rays, 2) - xys) / img_size * 2         # diff_flo, geom_utils.py:1745-1757
            valid = torch.ones_like(flo[..., :1])
        result[key + '_coarse'] = flo
        result[key + '_valid'] = valid
        flo_out[key] = (flo, valid)
    if 'nerf_unc' in models.keys():                                            # :501-516
        xyt = torch.cat([L.dev(rays['xysn']), L.dev(rays['ts'])], -1)
        result['unc_pred'] = models['nerf_unc'](torch.cat([embedding_xyz(xyt), L.dev(rays['vid_code'])], -1))
    sil_at = None
    if 'img_at_samp' in rays.keys():                                           # :518-571 (O(N) terms)
        img_at, sil_at, vis_at = (L.dev(rays[k]) for k in ('img_at_samp', 'sil_at_samp', 'vis_at_samp'))
        flo_at, cfd_at = L.dev(rays['flo_at_samp']), L.dev(rays['cfd_at_samp'])
        if 'flo' not in flo_out:
            raise KeyError("flo_coarse")   # the reference needs rtk_vec_target here too (rendering.py:549)
        flo, valid = flo_out['flo']
        # the reference branches on `.sum() > 0` tests and boolean-mask gathers here (rendering.py:535-539, 554-555: host
        # syncs, SURVEY 8a note 10) over ~25 eager ops; the three terms and the batch statistics they need are ONE kernel each
        # way (autograd.RayLossFn / moda_ray_loss), nothing leaves the device and the step stays graph-capturable
        s3im = bool(getattr(opts, 's3im_loss', False))
        img_obs = img_at.clone() if s3im else img_at          # (img_at itself is masked in place below when s3im_loss is on)
        img_loss_s, sil_loss, flo_loss_s, sil_flo = A.RayLossFn.apply(rgb, sil, flo, valid, img_obs, sil_at, vis_at, flo_at,
                                                                     cfd_at, is_training)
        if s3im:                                                               # :528-532, 566-567
            result['s3im_loss'] = LU.s3im_loss(rgb, img_obs, sil_at, rng=rng)
            # S3IM.forward multiplies its arguments by the mask IN PLACE (loss_utils.py:665-666), after img_loss_samp was formed:
            # with the flag on, the reference's result['img_coarse'] (the same tensor object as rgb_coarse, :402) and the
            # caller's rays['img_at_samp'] come back masked.  Mirrored: a masked img_coarse, and the observed colours masked
            # in place when they are the caller's own device tensor.
            result['img_coarse'] = rgb * sil_at
            with torch.no_grad():
                img_at.mul_(sil_at)
        result['img_at_samp'], result['sil_at_samp'], result['vis_at_samp'] = img_at, sil_at, vis_at
        result['sil_at_samp_flo'], result['flo_at_samp'] = sil_flo, flo_at
        result['img_loss_samp'] = img_loss_s
        result['sil_loss_samp'] = sil_loss
        result['flo_loss_samp'] = flo_loss_s
    if feats_at is not None:                                                   # :573-578
        if sil_at is None:
            raise KeyError("sil_at_samp")   # the reference reads it from the img_at_samp block
        frnd = A.RowDistFn.apply(A.NormalizeFn.apply(feat_rnd), feats_at, True)       # (normalize(feat_rnd) - feats_at).pow(2).mean(-1)
        result['frnd_loss_samp'] = frnd * sil_at[..., 0]


def _heads_present(rays):
    """Do the loss / correspondence heads behind compositing run (they read the (N, S) weights)?"""
    return any(k in rays.keys() for k in ('feats_at_samp', 'rtk_vec_target', 'rtk_vec_dentrg', 'bone_rts_target', 'bone_rts_dentrg',
                                          'img_at_samp'))


def _wants_grad(models, rays):
    if not torch.is_grad_enabled():
        return False
    for v in list(models.values()) + list(rays.values()):
        if torch.is_tensor(v) and v.requires_grad:
            return True
        if isinstance(v, torch.nn.Module) and any(p.requires_grad for p in v.parameters()):
            return True
    return False


def _inference_deform_train(xyz, rays, models, N_samples, N_rays, embedding_xyz, rays_d, noise_std, obj_bound,
                            dir_embedded, z_vals, opts, fine_iter, render_vis, rng, _pre, img_size=None):
    """Training route of inference_deform (rendering.py:239-579): same dataflow, every heavy node an autograd
    Function over the HIP kernels (moda_amd/autograd.py), activations kept for the backward (exact fp32)."""
    from . import autograd as A
    # every tensor below that feeds several autograd nodes is handed to each of them as its own alias (A.fanned): its gradient
    # is then ONE sum launch instead of one `add` per extra consumer (the warped positions alone feed eight nodes)
    with A.fan_scope():
        return _inference_deform_train_body(xyz, rays, models, N_samples, N_rays, embedding_xyz, rays_d, noise_std, obj_bound,
                                            dir_embedded, z_vals, opts, fine_iter, render_vis, rng, _pre, img_size)


def _inference_deform_train_body(xyz, rays, models, N_samples, N_rays, embedding_xyz, rays_d, noise_std, obj_bound,
                                 dir_embedded, z_vals, opts, fine_iter, render_vis, rng, _pre, img_size):
    from . import autograd as A
    F = A.fanned
    result = {}
    xyz_frame = xyz
    cyc = None
    dskin_f = None
    pts_tf = None
    has_bones = 'bones' in models.keys()
    emb = embedding_xyz   # dispatches to EmbedFn when its input carries a gradient
    if has_bones:
        bones_rst = L.dev(models['bones_rst'])
        B = bones_rst.shape[-2]
        rts = L.dev(rays['bone_rts']).reshape(N_rays, B, 8)
        skin_aux = L.dev(models['skin_aux'])
        nerf_skin = models['nerf_skin'] if 'nerf_skin' in models.keys() else None

        def dskin_of(pts, code):
            if nerf_skin is None:
                return None
            return nerf_skin.train_forward(pts, emb, code=code)                               # geom_utils.py:33-44

        bones_dfm = A.bone_transform(F(bones_rst).reshape(B, 10), F(rts))                   # rendering.py:303
        dskin = dskin_of(F(xyz_frame), L.dev(rays['time_embedded']).reshape(N_rays, -1))     # :304
        xyz, _, _ = A.WarpFn.apply(A.bone_prep(bones_dfm), A.dq_inverse(F(rts)), F(xyz_frame), dskin, F(skin_aux), None)   # :319
        nerf_dis = models['nerf_dis'] if 'nerf_dis' in models.keys() else None
        if nerf_dis is not None:                                                            # geom_utils.py:416-418
            xyz_dis = nerf_dis.train_forward(F(xyz_frame), emb, code=L.dev(rays['time_embedded']).reshape(N_rays, -1))
            xyz = xyz - xyz_dis
            result['dis_reg'] = xyz_dis.norm(dim=2)                                         # :321-322
        if fine_iter:
            rest = F(models['rest_pose_code'].weight).reshape(1, -1)
            dskin_f = dskin_of(F(xyz), rest)                                                # :330
            if nerf_dis is not None:                                                        # geom_utils.py:420-425
                dis_f = nerf_dis.train_forward(F(xyz), emb, code=F(models['rest_pose_code'].weight).reshape(1, -1))
                pts_tf = F(xyz) + dis_f
                result['dis_reg_forward'] = dis_f.norm(dim=2)                               # :342-343
            _, cyc, _ = A.WarpFn.apply(A.bone_prep(F(bones_rst).reshape(1, B, 10)), F(rts), F(xyz), F(dskin_f), F(skin_aux),
                                       F(xyz_frame), pts_tf)                                 # :338-341
    clip_bound, vis_pred = None, None
    if render_vis:
        with torch.no_grad():
            vis_pred = models['nerf_vis'].fused(xyz.detach(), n_freq=embedding_xyz.N_freqs, alpha=embedding_xyz.alpha,
                                                with_sigma=False, sigmoid=True, precision="fp32")[..., 0].contiguous()
        ob = tuple(float(b) for b in torch.as_tensor(obj_bound).reshape(-1)[:3].tolist())
        clip_bound = L.const_tensor(("bound", ob), xyz.device, lambda: torch.tensor(ob))
    xyz_in = xyz                           # (consumers below take F(xyz_in): aliases of the warped positions' one fan-out node)
    if opts.symm_shape:                                                                       # :385-391
        r = _draw(rng, 'symm_rand_pre' if _pre else 'symm_rand', "rand", (N_rays, N_samples, 1), xyz.device)
        xs = F(xyz)
        xyz_in = torch.cat([torch.where(r < 0.5, -xs[..., :1], xs[..., :1]), xs[..., 1:3]], -1)
    side = [dir_embedded.reshape(N_rays, -1)]                                                  # geom_utils.py:33-50 order
    if 'env_code' in rays.keys():
        side.append(L.dev(rays['env_code']).reshape(N_rays, -1))
    if 'appearance_code' in rays.keys():
        side.append(L.dev(rays['appearance_code']).reshape(N_rays, -1))
    rgbsigma = models['coarse'].train_forward(F(xyz_in), emb, dir_src=torch.cat(side, -1))    # :159
    # random draws in the reference's order (SURVEY 8a note 9): the (N,S) density noise (:193) comes before feat_match's lattice
    # jitter (loss_utils.py:306)
    noise_raw = (rng or {}).get('noise_raw_pre' if _pre else 'noise_raw')
    if noise_raw is None:
        noise_raw = torch.randn((N_rays, N_samples), device=xyz.device)                       # :193
    noise = None if noise_std == 0 else L.dev(noise_raw).reshape(N_rays, N_samples) * noise_std
    feat, feat_grid = None, None
    if 'nerf_feat' in models.keys():                                                           # :174-178
        if fine_iter and 'feats_at_samp' in rays.keys():
            # feat_match (rendering.py:417-437 -> loss_utils.py:300-313) evaluates the same network on its 20^3 lattice: both
            # point sets go through ONE call (one launch chain and one set of weight-gradient GEMMs instead of two)
            from . import loss_utils as LU
            q = LU.feat_grid_query(obj_bound, xyz.device, 20, models['coarse'].training, rng)
            both = models['nerf_feat'].train_forward(torch.cat([F(xyz_in).reshape(-1, 3), q], 0), emb)
            n_s = N_rays * N_samples
            f_s, f_g = A.SplitRowsFn.apply(both, n_s)
            feat = f_s.reshape(N_rays, N_samples, -1)
            feat_grid = (q, f_g)
        else:
            feat = models['nerf_feat'].train_forward(F(xyz_in), emb)
    rgb, feat_o, depth, sil, weights, vis, vis_o, cyc_o = A.CompositeFn.apply(
        rgbsigma, feat, z_vals, rays_d, models['coarse'].beta, noise, xyz_in, clip_bound, vis_pred,
        cyc if fine_iter else None, float(opts.scale_rgb) if getattr(opts, 'rgb_filter', False) else 0.0)
    result['img_coarse'] = rgb
    result['depth_rnd'] = depth
    result['sil_coarse'] = sil
    if render_vis:
        result['vis_pred'] = vis_o
    if fine_iter:
        result['xyz_camera_vis'] = xyz_frame
        if has_bones:
            result['xyz_canonical_vis'] = xyz
            result['frame_cyc_dis'] = cyc_o
        if feat is not None:
            result['feat_rnd'] = feat_o

        _corresp_and_loss_heads(result, rays, models, opts, img_size, weights, xyz, rgb, sil, embedding_xyz=embedding_xyz,
                                obj_bound=obj_bound, vis=vis, feat_rnd=feat_o, chunk=None, rng=rng, dskin_rest=dskin_f,
                                pts_tf=pts_tf, feat_grid=feat_grid)
    return result, weights


def inference_deform(xyz_coarse_sampled, rays, models, chunk, N_samples, N_rays, embedding_xyz, rays_d, noise_std,
                     obj_bound, dir_embedded, z_vals, img_size, progress, opts, fine_iter=True, render_vis=False,
                     rng=None, _pre=False, n_live=None, term_tau=0.0, _runs=None, _want_weights=True, _keep=False, _reuse=None):
    """rendering.py:239-579 (bones / neudbs and plain-NeRF branches) -> (result dict, weights).
    n_live / term_tau: opt-in early ray termination of the inference route (see render_rays).  _runs: the joint run partition of
    the per-ray rows when render_rays has computed it already; _want_weights=False (render_rays' final pass without loss heads):
    the (N, S) compositing weights are not written and None is returned in their place."""
    if 'flowbw' in models.keys():
        raise NotImplementedError("flowbw/flowfw free-form deformation is not MoDA's configuration (moda.py:72-73)")
    if getattr(opts, 'lbs', False):
        raise NotImplementedError("linear blend skinning: MoDA runs neudbs (moda.py:72-73)")
    if _wants_grad(models, rays) or (torch.is_grad_enabled() and xyz_coarse_sampled.requires_grad):
        if term_tau > 0 or n_live is not None:
            raise NotImplementedError("early ray termination is an inference-only option: the training route composes every "
                                      "sample, as the reference does (rendering.py:217-221)")
        return _inference_deform_train(xyz_coarse_sampled, rays, models, N_samples, N_rays, embedding_xyz, rays_d,
                                       noise_std, obj_bound, dir_embedded, z_vals, opts, fine_iter, render_vis, rng, _pre,
                                       img_size=img_size)
    nf, alpha = embedding_xyz.N_freqs, embedding_xyz.alpha
    xyz_frame = L.dev(xyz_coarse_sampled)                                      # :255 clone not needed: never mutated
    xyz = xyz_frame
    # REUSE_COARSE (see there).  _keep: this is the pre-pass and the final pass will reuse it -- full network outputs, no feature
    # net, result['_xyz_canon'] / ['_rgbsigma'].  _reuse: this is the final pass and xyz_coarse_sampled holds the IMPORTANCE points
    # only, (N, S_f, 3); the backward warp runs on them, then both halves are put in depth order (N_samples = the merged count)
    S_warp = xyz_frame.shape[1] if _reuse is not None else N_samples
    result = {}
    cyc = None
    dskin_f = None
    pts_tf = None
    has_bones = 'bones' in models.keys()
    if has_bones:
        bones_rst = models['bones_rst']                                        # :290
        bone_rts_fw = rays['bone_rts']
        skin_aux = models['skin_aux']
        nerf_skin = models['nerf_skin'] if 'nerf_skin' in models.keys() else None
        time_embedded = rays['time_embedded']                                  # (N,128); [:,None] in the reference
        if not getattr(opts, 'neudbs', True):
            raise NotImplementedError("opts.neudbs must be set (moda.py:72-73)")
        # rows of bone_rts / time_embedded: one per ray, or one per frame in the frame-grouped layout (FRAME_KEYS)
        n_sets = L.dev(bone_rts_fw).reshape(-1, bone_rts_fw.shape[-1]).shape[0]
        rps = N_rays // n_sets
        dskin = None
        # throughput mode: skin MLP -> skinning softmax -> DQS in ONE kernel per warp, the (N,B,S) logits stay in registers
        warp_prec = WARP_PRECISION.get(get_precision())
        if _pre and get_precision() == "fp16" and FP16_PREPASS_WARP:
            warp_prec = FP16_PREPASS_WARP
        if _pre and _PREPASS_OF_FP16 and get_precision() == "bf16x3" and FP16_X3_PREPASS_WARP:
            warp_prec = FP16_X3_PREPASS_WARP
        one_kernel = nerf_skin is not None and warp_prec is not None and FUSED_WARP
        # The reference's layout repeats every frame's rows per ray (moda.py:1302-1310): with many sets, the runs of identical
        # (bone_rts, time_embedded) rows are detected ONCE per call, on the device, and everything per-frame below -- bone_transform,
        # the code folds, the operand tables of both warps -- runs at the run starts only
        runs = None
        te_rows = L.dev(time_embedded).reshape(-1, time_embedded.shape[-1])
        if (one_kernel and n_sets >= 512 and te_rows.shape[0] == n_sets and ROW_RUNS
                and nerf_skin.fused_warp_serves(S_warp, embedding_xyz, warp_prec)
                and (_reuse is None or nerf_skin.fused_warp_serves(N_samples, embedding_xyz, warp_prec))):
            runs = _runs if _runs is not None else joint_row_runs(L.dev(bone_rts_fw).reshape(n_sets, -1), te_rows)
        bones_dfm = bone_transform(bones_rst, bone_rts_fw, True, is_vec=True, run_start=runs)  # :303
        done = None
        if one_kernel:
            done = nerf_skin.fused_warp(xyz, embedding_xyz, time_embedded, bones_dfm, bone_rts_fw, skin_aux, backward=True,
                                        rays_per_set=rps, precision=warp_prec, runs=runs, runs_cover_code=runs is not None)   # :304-319
            if done is None and runs is not None:       # (cannot happen: fused_warp_serves said yes) -- the other route needs every row
                bones_dfm = bone_transform(bones_rst, bone_rts_fw, True, is_vec=True)
        if done is not None:
            xyz = done[0]
        else:
            if nerf_skin is not None:                                          # :304 gauss_mlp_skinning
                # (N,B,S) layout: consecutive samples contiguous, so both this store and the warp's loads coalesce
                dskin = nerf_skin.fused(xyz, n_freq=nf, alpha=alpha,
                                        code=L.dev(time_embedded).reshape(-1, time_embedded.shape[-1]), out_tr_S=S_warp)
            xyz, _, _ = warp(bones_dfm, bone_rts_fw, xyz, dskin, skin_aux, backward=True, dskin_bns=True,
                             rays_per_set=rps)                                 # :319
        if _reuse is not None:                 # both halves in depth order: observed-frame points and their canonical images
            _reuse = dict(_reuse, canon_fine=xyz)
            xyz_frame = _merge_rows(_reuse['src'], _reuse['frame'], xyz_frame)
            xyz = _merge_rows(_reuse['src'], _reuse['canon'], xyz)
        nerf_dis = models['nerf_dis'] if 'nerf_dis' in models.keys() else None  # :307-310 residual displacement field
        if nerf_dis is not None:                                               # geom_utils.py:416-418: x* = DQS(x) - dis(x, t)
            xyz_dis = nerf_dis.fused(xyz_frame, n_freq=nf, alpha=alpha,
                                     code=L.dev(time_embedded).reshape(-1, time_embedded.shape[-1]))
            xyz = xyz - xyz_dis
            result['dis_reg'] = xyz_dis.norm(dim=2)                            # :321-322
        if fine_iter:
            rest = models['rest_pose_code'].weight                              # Embedding(1,128) row 0 (:293-294)
            if nerf_dis is not None:                                           # geom_utils.py:420-425: DQS of x* + dis(x*, rest)
                dis_f = nerf_dis.fused(xyz, n_freq=nf, alpha=alpha, code=L.dev(rest).reshape(1, -1))
                pts_tf = xyz + dis_f
                result['dis_reg_forward'] = dis_f.norm(dim=2)                  # :342-343
            # the loss heads re-use the rest-pose logits (dskin_f) for the target-frame warps: keep the two-kernel route then
            heads_need_dskin = any(kk in rays.keys() for kk in ('bone_rts_target', 'bone_rts_dentrg'))
            done = None
            if one_kernel and not heads_need_dskin:
                done = nerf_skin.fused_warp(xyz, embedding_xyz, rest, bones_rst, bone_rts_fw, skin_aux, backward=False,
                                            rays_per_set=rps, pts_tf=pts_tf, cyc_ref=xyz_frame, precision=warp_prec, runs=runs,
                                            want_xyz=False)                    # :330-341 (only the cycle distance is used, :341)
            if done is not None:
                cyc = done[1]
            else:
                if nerf_skin is not None:                                      # :330
                    dskin_f = nerf_skin.fused(xyz, n_freq=nf, alpha=alpha, code=L.dev(rest).reshape(1, -1),
                                              out_tr_S=N_samples)
                _, _, cyc = warp(bones_rst, bone_rts_fw, xyz, dskin_f, skin_aux, backward=False,
                                 cyc_ref=xyz_frame, dskin_bns=True, rays_per_set=rps, pts_tf=pts_tf)   # :338-341
    if _reuse is not None and not has_bones:   # plain NeRF: no warp, the canonical points are the frame points
        _reuse = dict(_reuse, canon_fine=xyz)
        xyz_frame = _merge_rows(_reuse['src'], _reuse['frame'], xyz_frame)
        xyz = xyz_frame
    env_code = rays['env_code'] if 'env_code' in rays.keys() else None         # :364-372
    appearance_code = rays['appearance_code'] if 'appearance_code' in rays.keys() else None
    clip_bound, vis_pred = None, None
    if render_vis:                                                             # :375-379
        clip_bound = obj_bound
        vis_pred = models['nerf_vis'].fused(xyz, n_freq=nf, alpha=alpha, with_sigma=False, sigmoid=True)[..., 0]
        vis_pred = vis_pred.contiguous()
    flip = None
    if opts.symm_shape:                                                        # :385-391
        r = _draw(rng, 'symm_rand_pre' if _pre else 'symm_rand', "rand", (N_rays, N_samples, 1), xyz.device)
        flip = (r < 0.5).reshape(N_rays, N_samples)
    # the coarse pre-pass of hierarchical sampling keeps only its weights (rendering.py:96-106: `_, weights_coarse = ...`),
    # which depend on the density alone: colour branch and feature net are dead work there (SURVEY 8a note 11)
    o = inference(models, embedding_xyz, xyz, rays_d, dir_embedded, z_vals, N_rays, N_samples, chunk, noise_std,
                  weights_only=bool(_pre) and not fine_iter and not _keep, env_code=env_code, appearance_code=appearance_code, clip_bound=clip_bound,
                  vis_pred=vis_pred, scale_rgb=opts.scale_rgb, rgb_filter=opts.rgb_filter, flip=flip,
                  noise_raw=(rng or {}).get('noise_raw_pre' if _pre else 'noise_raw'),
                  cyc=cyc if fine_iter else None, _full=True, n_live=n_live, term_tau=term_tau,
                  _want_visibility=fine_iter and models['coarse'].training and 'nerf_vis' in models.keys(),   # :395 (:224 feeds :475-477 only)
                  _want_weights=_want_weights or _heads_present(rays), _feat_consumed='feats_at_samp' in rays.keys(),
                  _keep=_keep, _reuse=_reuse)
    weights = o["weights"]
    if _keep:
        result['_xyz_canon'], result['_rgbsigma'] = xyz, o["_rgbsigma"]
    if _pre and has_bones:
        result['_runs'] = runs          # (the same rays in the final pass: the same partition)
    if o["n_used"] is not None:
        result['samples_used'] = o["n_used"]       # not a reference key: present only with early termination switched on
    result['img_coarse'] = o["rgb"]                                            # :402-404
    result['depth_rnd'] = o["depth"]
    result['sil_coarse'] = o["sil"]
    if render_vis:
        result['vis_pred'] = o["vis_out"]                                      # :408
    if fine_iter:
        result['xyz_camera_vis'] = xyz_frame                                   # :464
        if has_bones:
            result['xyz_canonical_vis'] = xyz                                  # :466
            result['frame_cyc_dis'] = o["cyc_out"]                             # :473
        if 'nerf_feat' in models.keys():
            result['feat_rnd'] = o["feat"]     # not a reference key: rendered features, exposed for inspection

        _corresp_and_loss_heads(result, rays, models, opts, img_size, weights, xyz, o["rgb"], o["sil"],
                                embedding_xyz=embedding_xyz, obj_bound=obj_bound, vis=o["visibility"], feat_rnd=o["feat"],
                                chunk=chunk, rng=rng, dskin_rest=dskin_f, dskin_bns=True, pts_tf=pts_tf)
    return result, weights


def render_rays(models, embeddings, rays, N_samples=64, use_disp=False, perturb=0, noise_std=1, chunk=1024 * 32,
                obj_bound=None, use_fine=False, img_size=None, progress=None, opts=None, render_vis=False, rng=None):
    """rendering.py:19-122.  Same arguments and result keys; `rng` optionally injects the random draws."""
    if use_fine:
        N_samples = N_samples // 2                                             # :50
    embedding_xyz = embeddings['xyz']
    embedding_dir = embeddings['dir']
    rays_o = L.dev(rays['rays_o'])
    rays_d = L.dev(rays['rays_d'])
    near = L.dev(rays['near']).reshape(-1)
    far = L.dev(rays['far']).reshape(-1)
    train = _wants_grad(models, rays)
    N_rays = rays_d.shape[0]
    device = rays_d.device
    rays, _ = _frame_layout(rays, N_rays, train)
    if N_rays == 0:      # an empty shard: nothing to launch, the keys of a plain call with empty tensors
        S_out = 2 * N_samples if use_fine else N_samples
        e = lambda *shape: torch.zeros(shape, device=device)
        result = {'img_coarse': e(0, 3), 'depth_rnd': e(0), 'sil_coarse': e(0), 'xyz_camera_vis': e(0, S_out, 3)}
        if 'bones' in models.keys():
            result['xyz_canonical_vis'] = e(0, S_out, 3)
            result['frame_cyc_dis'] = e(0)
        return result
    dir_embedded = embedding_dir(rays_d, normalize=True)                       # :64-65 (EmbedFn when rays_d needs grad)
    u = None
    if perturb > 0:
        u = _draw(rng, 'perturb_rand', "rand", (N_rays, N_samples), device)    # :82
    z_vals = torch.empty((N_rays, N_samples), device=device)
    xyz = torch.empty((N_rays, N_samples, 3), device=device)
    L.call("moda_sample_rays_fwd", L.ptr(rays_o), L.ptr(rays_d), L.ptr(near), L.ptr(far), L.ptr(u), float(perturb),
           int(bool(use_disp)), N_rays, N_samples, L.ptr(z_vals), L.ptr(xyz), L.stream())   # :68-89
    if train and not use_fine:
        # depths are not optimised ("zvals are not optimized", rendering.py:85): gradients reach the rays through xyz
        from .autograd import PointsFn
        xyz = PointsFn.apply(rays_o, rays_d, z_vals)
    # Early ray termination -- opt-in through opts.early_term_tau (default 0 = the reference's arithmetic, which composes
    # every sample, rendering.py:217-221); inference route only.  The final compositing drops the samples whose incoming
    # transmittance is below tau (at most tau of a ray's weight).  With hierarchical sampling the coarse pre-pass also
    # yields each ray's termination depth (at tau / 10, a margin for its coarser quadrature) and the 8x256 MLP of the final
    # pass skips the 32-sample groups behind it.
    tau = float(getattr(opts, 'early_term_tau', 0.0) or 0.0)
    if tau < 0 or tau >= 1:
        raise ValueError(f"opts.early_term_tau={tau}: expected a transmittance threshold in [0, 1)")
    n_live = None
    if use_fine:                                                               # :91-114
        # fp16 mode: the pre-pass decides where the second half of the samples goes, and the inverse CDF of sample_pdf divides a
        # weight error by the bin's probability (:612-621) -- depths moved by 1e-3 with fp16 operands (G7 `fine_perturb_symm`).
        # It runs split-bf16 (3 MFMAs per product on half of the samples, sigma only); the final pass keeps fp16.
        global _PREPASS_OF_FP16
        outer_fp16 = get_precision() == "fp16" and not train
        reuse = (REUSE_COARSE and not train and tau == 0 and not opts.symm_shape and 'nerf_dis' not in models.keys()
                 and 'flowbw' not in models.keys() and not getattr(opts, 'lbs', False))
        with torch.no_grad(), precision_scope(FP16_PREPASS_PRECISION if outer_fp16 else None):   # :96
            _PREPASS_OF_FP16 = outer_fp16
            try:
                pre, w = inference_deform(xyz, rays, models, chunk, N_samples, N_rays, embedding_xyz, rays_d, noise_std,
                                          obj_bound, dir_embedded.detach(), z_vals, img_size, progress, opts,
                                          fine_iter=False, rng=rng, _pre=True, term_tau=0.1 * tau if not train else 0.0,
                                          _keep=reuse)
            finally:
                _PREPASS_OF_FP16 = False
        z_term = None
        if tau > 0 and not train:
            used = pre['samples_used'].long()                                  # first coarse sample with T < tau / 10, or S
            z_term = torch.where(used < N_samples, z_vals.gather(1, used.clamp(max=N_samples - 1)[:, None])[:, 0],
                                 torch.full_like(z_vals[:, 0], float('inf')))
        z_mid = 0.5 * (z_vals[:, :-1] + z_vals[:, 1:])                         # :105
        pu = None
        if perturb != 0:
            pu = _draw(rng, 'pdf_u', "rand", (N_rays, N_samples), device)      # :607
        z_new = sample_pdf(z_mid.contiguous(), w[:, 1:-1].contiguous(), N_samples, det=(perturb == 0), u=pu)   # :106
        if reuse:
            # the merged depths with their origin; the final pass below evaluates the importance points only (REUSE_COARSE)
            z_vals, src = _merge_index(z_vals, z_new)                          # :110
            xyz_fine = torch.empty((N_rays, N_samples, 3), device=device)
            L.call("moda_points_fwd", L.ptr(rays_o), L.ptr(rays_d), L.ptr(z_new), N_rays, N_samples, L.ptr(xyz_fine), L.stream())
            result, _ = inference_deform(xyz_fine, rays, models, chunk, 2 * N_samples, N_rays, embedding_xyz, rays_d, noise_std,
                                         obj_bound, dir_embedded, z_vals, img_size, progress, opts, render_vis=render_vis,
                                         rng=rng, _want_weights=False, _runs=pre.get('_runs'),
                                         _reuse=dict(src=src, frame=xyz, canon=pre['_xyz_canon'], rgbsigma=pre['_rgbsigma']))
            return result
        z_vals = _merge_sorted(z_vals, z_new)                                  # :110
        N_samples = 2 * N_samples                                              # :114
        if z_term is not None:
            n_live = (z_vals <= z_term[:, None]).sum(1).to(torch.int32)        # merged samples in front of the termination depth
        xyz = torch.empty((N_rays, N_samples, 3), device=device)
        if train:
            from .autograd import PointsFn
            xyz = PointsFn.apply(rays_o, rays_d, z_vals)
        else:
            L.call("moda_points_fwd", L.ptr(rays_o), L.ptr(rays_d), L.ptr(z_vals), N_rays, N_samples, L.ptr(xyz),
                   L.stream())                                                 # :112-113
    result, _ = inference_deform(xyz, rays, models, chunk, N_samples, N_rays, embedding_xyz, rays_d, noise_std,
                                 obj_bound, dir_embedded, z_vals, img_size, progress, opts, render_vis=render_vis,
                                 rng=rng, n_live=n_live, term_tau=tau, _want_weights=False,         # :116 (weights discarded here)
                                 _runs=pre.get('_runs') if (use_fine and not train) else None)
    return result
