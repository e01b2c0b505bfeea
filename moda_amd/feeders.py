"""The per-frame feeders immediately before the rendering path, with the reference's names and
argument meaning: ray construction (`raycast`, `sample_xy`, `chunk_rays`, nnutils/geom_utils.py:746-838), the frame codes
(`FrameCode`, nnutils/nerf.py:346-380), the body-pose head (`DQ_RTHead`, nerf.py:239-279), the rest-pose correction
(`correct_bones`, `correct_rest_pose`, geom_utils.py:933-972, composed by `update_delta_rts`, moda.py:1262-1279) and the
per-ray expansion `update_rays` performs (nnutils/moda.py:1281-1327).  Arithmetic runs in HIP kernels behind autograd Functions; gradients reach the camera
(`Rmat`, `Tmat`, `Kinv`), the code tables and the pose head's parameters as they do in the reference."""
import numpy as np
import torch
from torch import nn
from torch.autograd import Function

from . import _lib as L
from . import autograd as A
from .dual_quat import dq_inverse, dq_mul
from .geom_utils import bone_transform
from .nerf import Embedding, NeRF


class RaycastFn(Function):
    """geom_utils.py:763-766: rays_d = (Kinv [x,y,1])^T R, rays_o = -T^T R."""

    @staticmethod
    def forward(ctx, xys, Rmat, Tmat, Kinv):
        xy, R, T, K = (L.dev(t) for t in (xys, Rmat, Tmat, Kinv))
        bs, ns, _ = xy.shape
        d = torch.empty((bs, ns, 3), device=xy.device)
        o = torch.empty((bs, ns, 3), device=xy.device)
        L.call("moda_raycast", L.ptr(xy), L.ptr(R), L.ptr(T), L.ptr(K), bs, ns, L.ptr(d), L.ptr(o), None, None, None, None,
               None, L.stream())
        ctx.save_for_backward(xy, R, T, K)
        return d, o

    @staticmethod
    def backward(ctx, g_d, g_o):
        xy, R, T, K = ctx.saved_tensors
        bs, ns, _ = xy.shape
        gd = torch.zeros((bs, ns, 3), device=xy.device) if g_d is None else L.dev(g_d)
        go = None if g_o is None else L.dev(g_o)
        dR, dT, dK = torch.empty_like(R), torch.empty_like(T), torch.empty_like(K)
        L.call("moda_raycast", L.ptr(xy), L.ptr(R), L.ptr(T), L.ptr(K), bs, ns, None, None, L.ptr(gd), L.ptr(go), L.ptr(dR),
               L.ptr(dT), L.ptr(dK), L.stream())
        return None, dR, dT, dK


def raycast(xys, Rmat, Tmat, Kinv, near_far):
    """geom_utils.py:746-794 -> rays dict (rays_o, rays_d, near, far, rtk_vec, xys, nsample, bs), tensors (bs, ns, .)."""
    xys = L.dev(xys)
    bs, nsample, _ = xys.shape
    Rmat = L.dev(Rmat).reshape(-1, 3, 3)
    Tmat = L.dev(Tmat).reshape(-1, 3)
    Kinv = L.dev(Kinv).reshape(-1, 3, 3)
    rays_d, rays_o = RaycastFn.apply(xys, Rmat, Tmat, Kinv)
    if near_far is not None:
        nf = L.dev(near_far)
        znear = nf[:, 0, None, None].expand(bs, nsample, 1).contiguous()           # :769-770
        zfar = nf[:, 1, None, None].expand(bs, nsample, 1).contiguous()
    else:                                                                         # :772-776
        z = Tmat[:, None, 2:3].expand(bs, nsample, 1)
        znear = (z - 1.5).clamp_min(1e-5)
        zfar = z + 1.5
    rtk_vec = torch.cat([Rmat.reshape(-1, 1, 9), Tmat.reshape(-1, 1, 3), Kinv.reshape(-1, 1, 9)], -1)   # :780-784
    return {'rays_o': rays_o, 'rays_d': rays_d, 'near': znear, 'far': zfar,
            'rtk_vec': rtk_vec.expand(bs, nsample, 21).contiguous(), 'xys': xys, 'nsample': nsample, 'bs': bs}


def sample_xy(img_size, bs, nsample, device, return_all=False, lineid=None):
    """geom_utils.py:796-827: pixel indices and coordinates (index logic and torch's own sampler; no arithmetic)."""
    ar = torch.arange(img_size, device=device, dtype=torch.float32)
    xygrid = torch.stack([ar[None, :].expand(img_size, img_size), ar[:, None].expand(img_size, img_size)], -1).reshape(1, -1, 2)
    if return_all:
        xys = xygrid.repeat(bs, 1, 1)
        rand_inds = torch.arange(xys.shape[1], dtype=torch.float32)[None].repeat(bs, 1)
    elif lineid is None:
        rand_inds = torch.multinomial(torch.ones(img_size ** 2, device=device), bs * nsample, replacement=False).view(bs, nsample)
        xys = xygrid[0][rand_inds]
    else:
        rand_inds = torch.multinomial(torch.ones(img_size, device=device), bs * nsample, replacement=True).view(bs, nsample)
        xys = xygrid[0][rand_inds].clone()
        xys[..., 1] = xys[..., 1] + lineid[:, None]
    return rand_inds.long(), xys


def chunk_rays(rays, start, delta):
    """geom_utils.py:829-838: rays [start, start+delta) of the flattened (bs*nsample, C) tensors.  In the frame-grouped
    layout (`rays_per_frame` set by update_rays(frame_layout=True)) the rendering.FRAME_KEYS tensors hold one row per
    frame; the chunk must then cover whole frames and keeps the layout."""
    k = rays.get('rays_per_frame', None)
    if k is None:
        return {key: v.view(-1, v.shape[-1])[start:start + delta] for key, v in rays.items() if torch.is_tensor(v)}
    if start % k or delta % k:
        raise ValueError(f"frame-grouped rays: chunks must cover whole frames of {k} rays")
    from .rendering import FRAME_KEYS
    n_rays = rays['rays_d'].reshape(-1, 3).shape[0]
    out = {'rays_per_frame': k}
    for key, v in rays.items():
        if not torch.is_tensor(v):
            continue
        v2 = v.reshape(-1, v.shape[-1])
        if key in FRAME_KEYS and v2.shape[0] * k == n_rays:
            out[key] = v2[start // k:(start + delta) // k]
        else:
            out[key] = v2[start:start + delta]
    return out


def fid_reindex(fid, num_vids, vid_offset):
    """geom_utils.py:1759-1778: absolute frame id -> (video id, relative time in [-1, 1])."""
    vid_offset = np.asarray(vid_offset)
    tid = torch.zeros_like(fid).float()
    vid = torch.zeros_like(fid)
    max_ts = float((vid_offset[1:] - vid_offset[:-1]).max())
    for i in range(num_vids):
        assign = torch.logical_and(fid >= int(vid_offset[i]), fid < int(vid_offset[i + 1]))
        vid[assign] = i
        doffset = float(vid_offset[i + 1] - vid_offset[i])
        tid[assign] = (fid[assign].float() - float(vid_offset[i]) - doffset / 2) / max_ts * 2
    return vid, tid


class FrameCode(nn.Module):
    """nerf.py:346-380: frame index -> code = Linear(one-hot(video) (x) Fourier(t))."""

    def __init__(self, num_freq, embedding_dim, vid_offset, scale=1):
        super().__init__()
        self.vid_offset = np.asarray(vid_offset)
        self.num_vids = len(vid_offset) - 1
        max_ts = (self.vid_offset[1:] - self.vid_offset[:-1]).max()
        self.num_freq = 2 * int(np.log2(max_ts)) - 2
        self.fourier_embed = Embedding(1, num_freq, alpha=num_freq)
        self.basis_mlp = nn.Linear(self.num_vids * self.fourier_embed.out_channels, embedding_dim)
        self.scale = scale

    def forward(self, fid):
        bs = fid.shape[0]
        vid, tid = fid_reindex(fid, self.num_vids, self.vid_offset)
        coeff = self.fourier_embed(L.dev(tid * self.scale).view(bs, 1))                      # (bs, C), HIP
        C = coeff.shape[1]
        # coeff[..., None] * one_hot(vid): each row's C coefficients land in its video's column of a (C, num_vids) grid
        wide = torch.zeros((bs, C, self.num_vids), device=coeff.device)
        wide.scatter_(2, L.dev(vid, torch.int64).view(bs, 1, 1).expand(bs, C, 1), coeff[..., None])
        return A.LinearFn.apply(wide.view(bs, -1), self.basis_mlp.weight, self.basis_mlp.bias, 0)


class RtToDqFn(Function):
    @staticmethod
    def forward(ctx, rts):
        r = L.dev(rts)
        out = torch.empty((r.shape[0], 8), device=r.device)
        L.call("moda_rt_to_dq", L.ptr(r), r.shape[0], L.ptr(out), None, None, L.stream())
        ctx.save_for_backward(r)
        return out

    @staticmethod
    def backward(ctx, g):
        (r,) = ctx.saved_tensors
        d = torch.empty_like(r)
        L.call("moda_rt_to_dq", L.ptr(r), r.shape[0], None, L.ptr(L.dev(g)), L.ptr(d), L.stream())
        return d


class DQ_RTHead(NeRF):
    """nerf.py:239-279: code (bs, C) -> unit dual quaternions of the B bones, (bs, 1, 8B)."""

    def __init__(self, use_quat, **kwargs):
        super().__init__(**kwargs)
        if not use_quat:
            raise NotImplementedError("DQ_RTHead is built with use_quat=True (moda.py:314-319)")
        self.use_quat = use_quat
        self.num_output = 7
        for m in self.modules():
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()

    def forward(self, x):
        y = super().forward(x)
        bs = y.shape[0]
        return RtToDqFn.apply(y.reshape(-1, self.num_output)).view(bs, 1, -1)


def correct_bones(model, bones_rst, inverse=False, neudbs=True):
    """geom_utils.py:933-951: rest bones moved by the rest pose's transforms -> (bones_rst (B,10), bone_rts_rst (1, 8B))."""
    if not neudbs:
        raise NotImplementedError("linear blend skinning: MoDA runs neudbs (moda.py:72-73)")
    code = model.rest_pose_code.weight[:1]
    bone_rts_rst = model.nerf_body_rts[1](code)[0]
    B = bones_rst.shape[-2]
    if inverse:
        bone_rts_rst = dq_inverse(bone_rts_rst.view(-1, B, 8)).view(bone_rts_rst.shape)
    return bone_transform(bones_rst, bone_rts_rst, neudbs, is_vec=True)[0], bone_rts_rst


def correct_rest_pose(opts, bone_rts_fw, bone_rts_rst, neudbs):
    """geom_utils.py:953-972: delta(J_b) = (J_b*)^-1 J_b for every frame's bone transforms."""
    if not neudbs:
        raise NotImplementedError("linear blend skinning: MoDA runs neudbs (moda.py:72-73)")
    shape = bone_rts_fw.shape
    B = opts.num_bones
    inv = dq_inverse(bone_rts_rst.view(-1, B, 8))
    fw = bone_rts_fw.reshape(-1, B, 8)
    return dq_mul(inv.expand(fw.shape[0], B, 8).contiguous(), fw.contiguous()).view(shape)


def update_delta_rts(model, rays):
    """moda.update_delta_rts (moda.py:1262-1279): the rest bones moved by the rest pose (kept in nerf_models['bones_rst'])
    and every bone_rts* entry of `rays` re-expressed relative to the rest pose."""
    opts = model.opts
    bones_rst, bone_rts_rst = correct_bones(model, model.nerf_models['bones'], neudbs=opts.neudbs)
    model.nerf_models['bones_rst'] = bones_rst
    for k in ('bone_rts', 'bone_rts_target', 'bone_rts_dentrg'):
        if k in rays:
            rays[k] = correct_rest_pose(opts, rays[k], bone_rts_rst, opts.neudbs)
    return rays


def update_rays(model, rays, is_pair, embedid, frame_layout=False):
    """The per-ray expansion of moda.update_rays (moda.py:1281-1311) for the neudbs configuration: frame codes and body
    poses evaluated once per frame, then repeated over the frame's `nsample` pixels (the layout render_rays takes).
    frame_layout=True keeps them as ONE row per frame, (bs, C), and records rays['rays_per_frame'] = nsample: render_rays
    then reads 8B + 128 + 64 floats per frame instead of per ray (the frame-grouped layout, rendering.FRAME_KEYS)."""
    ns = rays['nsample']
    embedid = embedid.long()
    if frame_layout:
        rep = lambda t: t
        rays['rays_per_frame'] = ns
        rays['rtk_vec'] = rays['rtk_vec'][:, 0]
    else:
        rep = lambda t: t[:, None].expand(t.shape[0], ns, t.shape[-1])
    if is_pair:
        rv = rays['rtk_vec']
        rays['rtk_vec_target'] = rv.reshape((2, rv.shape[0] // 2) + tuple(rv.shape[1:])).flip(0).reshape(rv.shape)
        target = embedid.view(2, -1).flip(0).reshape(-1)
        rays['bone_rts_target'] = rep(model.nerf_body_rts(target)[:, 0])
    rays['time_embedded'] = rep(model.pose_code(embedid))
    rays['bone_rts'] = rep(model.nerf_body_rts(embedid)[:, 0])
    if getattr(model, 'env_code', None) is not None:
        rays['env_code'] = rep(model.env_code(embedid))
    if getattr(model, 'appearance_code', None) is not None:
        rays['appearance_code'] = rep(model.appearance_code(embedid))
    return rays
