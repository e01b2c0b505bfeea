"""Canonical-grid queries of mesh extraction: the point warps `warp_bw` / `warp_fw`
(nnutils/geom_utils.py:974-1073) and the volume queries `extract_mesh` runs before marching cubes
(nnutils/train_utils.py:1378-1422: `nerf_coarse(sigma_only=True)` and `sigmoid(nerf_vis)` on a grid^3 lattice).
Same kernels as the rendering path, driven by points instead of rays: all P points of a call share one frame, so they
are laid out as ONE ray of P samples -- the per-frame bone data is prepared once and read through wave-uniform loads.
Inference only (the reference runs these under no_grad)."""
import numpy as np
import torch

from . import _lib as L
from .feeders import correct_bones, correct_rest_pose
from .geom_utils import bone_transform, warp


def _frame_transforms(opts, model, embedid, device):
    if getattr(opts, 'flowbw', False) or getattr(opts, 'lbs', False) or not getattr(opts, 'neudbs', True):
        raise NotImplementedError("flowbw / lbs warps: MoDA runs neudbs (moda.py:72-73)")
    query_time = torch.full((1,), int(embedid), dtype=torch.long, device=device)
    bone_rts_fw = model.nerf_body_rts(query_time)                                     # (1,1,8B)   :991 / :1040
    bones_rst, bone_rts_rst = correct_bones(model, model.bones, neudbs=True)           # :993 / :1041
    bone_rts_fw = correct_rest_pose(opts, bone_rts_fw, bone_rts_rst, True)             # :994 / :1042
    return query_time, bones_rst, bone_rts_fw.reshape(1, -1)


@torch.no_grad()
def warp_bw(opts, model, rt_dict, query_xyz_chunk, embedid):
    """geom_utils.py:974-1027: observed-space points (P,3) of frame `embedid` -> canonical space."""
    pts = L.dev(query_xyz_chunk).reshape(1, -1, 3)
    P = pts.shape[1]
    query_time, bones_rst, rts = _frame_transforms(opts, model, embedid, pts.device)
    emb = model.embedding_xyz
    bones_dfm = bone_transform(bones_rst, rts, True, is_vec=True)                      # (1,B,10)  :1004
    dskin = None
    if getattr(opts, 'nerf_skin', True):
        time_embedded = model.pose_code(query_time)                                    # (1,128)   :1003
        dskin = model.nerf_skin.fused(pts, n_freq=emb.N_freqs, alpha=emb.alpha, code=L.dev(time_embedded).reshape(1, -1),
                                      out_tr_S=P)
    out = warp(bones_dfm, rts, pts, dskin, model.skin_aux, backward=True, dskin_bns=True)[0]   # :1006-1022
    if getattr(opts, 'nerf_dis', False):                                               # :1010-1022, geom_utils.py:416-418
        out = out - model.nerf_dis.fused(pts, n_freq=emb.N_freqs, alpha=emb.alpha,
                                         code=L.dev(model.pose_code(query_time)).reshape(1, -1))
    rt_dict['bones'] = bones_dfm                   # what neu_dbs returns as bones_dfm (geom_utils.py:387): per point identical
    return out.reshape(-1, 3), rt_dict


@torch.no_grad()
def warp_fw(opts, model, rt_dict, vertices, embedid):
    """geom_utils.py:1029-1073: canonical vertices (P,3) -> observed space of frame `embedid` (numpy out, as the reference)."""
    dev = L.dev(model.bones).device
    pts = torch.as_tensor(np.asarray(vertices), dtype=torch.float32, device=dev).reshape(1, -1, 3)
    P = pts.shape[1]
    _, bones_rst, rts = _frame_transforms(opts, model, embedid, dev)
    emb = model.embedding_xyz
    dskin = None
    if getattr(opts, 'nerf_skin', True):
        rest = model.rest_pose_code.weight[:1]                                          # :1049-1050
        dskin = model.nerf_skin.fused(pts, n_freq=emb.N_freqs, alpha=emb.alpha, code=L.dev(rest), out_tr_S=P)
    pts_tf = None
    if getattr(opts, 'nerf_dis', False):                                               # :1060-1069, geom_utils.py:420-422
        pts_tf = pts + model.nerf_dis.fused(pts, n_freq=emb.N_freqs, alpha=emb.alpha, code=L.dev(model.rest_pose_code.weight[:1]))
    out = warp(bones_rst, rts, pts, dskin, model.skin_aux, backward=False, dskin_bns=True, pts_tf=pts_tf)[0]   # :1052-1066
    rt_dict['bones'] = bone_transform(bones_rst, rts, True, is_vec=True)
    return out.reshape(-1, 3).cpu().numpy(), rt_dict


def query_grid(bound, grid_size):
    """train_utils.py:1378-1389: lattice points (x_i, y_j, z_k), C-order over (i, j, k), as a (grid^3, 3) array."""
    from .loss_utils import _query_grid       # the same expression as loss_utils.py:290-294, on the caller's scalars (dtype included)
    return _query_grid(bound, grid_size)


@torch.no_grad()
def query_volume(nerf_coarse, embedding_xyz, bound, grid_size, nerf_vis=None, point_warp=None, symm_shape=False,
                 precision=None):
    """The volume queries of extract_mesh (train_utils.py:1390-1422): vol_o (g,g,g) = nerf_coarse(sigma_only) on the
    lattice (optionally backward-warped first by `point_warp(points) -> points`, and folded onto x >= 0 for symm_shape),
    and vis (g,g,g) = sigmoid(nerf_vis) on the un-warped lattice, or None.  One fused launch each for the whole lattice
    (256^3 = 16.8 M points fit comfortably in HBM; the reference's `chunk` loop is not needed)."""
    dev = next(nerf_coarse.parameters()).device
    q = torch.from_numpy(query_grid(bound, grid_size)).to(dev)
    pts = q
    if point_warp is not None:
        pts = point_warp(q.clone())
    if symm_shape:
        pts = torch.cat([pts[:, :1].abs(), pts[:, 1:]], -1)                              # :1399-1400
    nf, alpha = embedding_xyz.N_freqs, embedding_xyz.alpha
    vol_o = nerf_coarse.fused(pts, n_freq=nf, alpha=alpha, sigma_only=True, precision=precision)
    vis = None
    if nerf_vis is not None:
        vis = nerf_vis.fused(q, n_freq=nf, alpha=alpha, with_sigma=False, sigmoid=True, precision=precision)[..., 0]
        vis = vis.reshape(grid_size, grid_size, grid_size)
    return vol_o.reshape(grid_size, grid_size, grid_size), vis
