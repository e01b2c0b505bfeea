"""Skinning subset of the reference's nnutils/geom_utils.py (same names, argument meaning and tensor
layouts), evaluated by the HIP library: evaluate_mlp :19-57, bone_transform :59-111 (neudbs branch),
vec_to_sim3 :187-199, gauss_mlp_skinning :202-217, mlp_skinning :219-229, skinning :280-302,
neu_dbs :372-456, dqs_blend_skinning :495-517.
"""
import torch

from . import _lib as L
from .nerf import Embedding, NeRF


def _per_row(t, n_rows):
    """(n,c), (n,1,c), (1,c), or an expanded (n,S,c) view -> a (R,c) tensor with R in {1, n}; else None."""
    if t.dim() == 3:
        if t.shape[1] == 1:
            return t[:, 0]
        if t.stride(1) == 0:
            return t[:, 0]
        return None
    return t


def evaluate_mlp(model, xyz_embedded, embed_xyz=None, dir_embedded=None, chunk=32 * 1024, xyz=None, code=None,
                 appearance_code=None, sigma_only=False, use_semantic=False):
    """geom_utils.py:19-57.  `chunk` only bounded the reference's memory and is not needed here.

    With `embed_xyz` given (raw sample positions in, the hot path) and per-ray side inputs, the whole
    embed -> concat -> MLP chain is one fused kernel; otherwise the inputs are concatenated as the
    reference does and the model's layer-by-layer route is used."""
    B, nbins, _ = xyz_embedded.shape
    fusable = isinstance(model, NeRF) and isinstance(embed_xyz, Embedding) and embed_xyz.in_channels == 3 \
        and xyz_embedded.shape[-1] == 3 and model.skips == [4] and 5 <= model.D <= 8 \
        and model.W in (64, 128, 256) and embed_xyz.N_freqs <= 10
    if fusable:
        d_rows = None if dir_embedded is None else _per_row(dir_embedded, B)
        c_rows = None if code is None else _per_row(code, B)
        a_rows = None if appearance_code is None else _per_row(appearance_code, B)
        fusable = (dir_embedded is None or d_rows is not None) and (code is None or c_rows is not None) \
            and (appearance_code is None or a_rows is not None)
    if fusable:
        # geom_utils.py:33-50 column order: [PE(xyz) | dir_embedded | code | appearance_code]; the part after
        # in_channels_xyz is the `input_dir` split of NeRF.forward (nerf.py:166-167)
        side = [t for t in (d_rows, c_rows, a_rows) if t is not None]
        n_pe = embed_xyz.out_channels
        code_in, dir_in = None, None
        if model.in_channels_xyz > n_pe:     # nerf_skin-style: code rides with the xyz input
            if dir_embedded is not None or len(side) != 1 or side[0].shape[-1] != model.in_channels_xyz - n_pe:
                fusable = False
            else:
                code_in = side[0]
        elif side:
            rows = max(t.shape[0] for t in side)
            side = [t if t.shape[0] == rows else t.expand(rows, t.shape[1]) for t in side]
            dir_in = torch.cat([L.dev(t) for t in side], -1)
            if dir_in.shape[-1] != model.in_channels_dir:
                fusable = False
        elif model.in_channels_dir != 0 and not sigma_only:
            fusable = False
    if fusable:
        return model.fused(xyz_embedded, n_freq=embed_xyz.N_freqs, alpha=embed_xyz.alpha, code=code_in,
                           dir_src=dir_in, sigma_only=sigma_only)
    # ---- general route: materialise the concatenation exactly as the reference does -----------------
    embedded = xyz_embedded
    if embed_xyz is not None:
        embedded = embed_xyz(embedded)
    if dir_embedded is not None:
        embedded = torch.cat([embedded, dir_embedded], -1)
    if code is not None:
        if code.shape[0] != B and code.dim() == 2:
            code = code.repeat(B, 1)
        if code.dim() == 2:
            code = code[:, None]
        embedded = torch.cat([embedded, code.expand(B, nbins, code.shape[-1])], -1)
    if appearance_code is not None:
        ac = appearance_code[:, None] if appearance_code.dim() == 2 else appearance_code
        embedded = torch.cat([embedded, ac.expand(B, nbins, ac.shape[-1])], -1)
    return model(embedded, sigma_only=sigma_only, xyz=xyz)


def _grad(*ts):
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in ts)


def bone_transform(bones_in, rts, neudbs, is_vec=False, run_start=None):
    """geom_utils.py:59-111: bones (..,B,10) rest Gaussians, rts (...,B*8) dual quaternions -> (bs,B,10).
    run_start (bs,) int32 (not a reference argument; inference route): only rows that start a run of identical `rts` rows are
    computed -- the others are left uninitialised, for consumers that read a run's first row."""
    if not neudbs:
        raise NotImplementedError("only the neudbs (dual-quaternion) branch is on MoDA's path (moda.py:72-73)")
    B = bones_in.shape[-2]
    if _grad(bones_in, rts):
        from . import autograd as A
        return A.bone_transform(L.dev(bones_in).reshape(B, 10), L.dev(rts).reshape(-1, B, 8))
    bones = L.dev(bones_in).reshape(-1, B, 10)
    if bones.shape[0] != 1:
        raise NotImplementedError("bone_transform expects one set of rest bones (B,10)")
    r = L.dev(rts).reshape(-1, B, 8)
    out = torch.empty((r.shape[0], B, 10), device=r.device, dtype=torch.float32)
    L.call("moda_bone_transform_fwd", L.ptr(bones), L.ptr(r), r.shape[0], B, L.ptr(out), L.ptr(run_start), L.stream())
    return out


def vec_to_sim3(vec):
    """geom_utils.py:187-199 -> center (...,3), orient (...,3,3), scale (...,3)."""
    L.no_grad_only(vec)
    lead = vec.shape[:-1]
    v = L.dev(vec).reshape(-1, 10)
    n = v.shape[0]
    c = torch.empty((n, 3), device=v.device)
    o = torch.empty((n, 9), device=v.device)
    s = torch.empty((n, 3), device=v.device)
    L.call("moda_vec_to_sim3_fwd", L.ptr(v), n, L.ptr(c), L.ptr(o), L.ptr(s), L.stream())
    return c.view(lead + (3,)), o.view(lead + (3, 3)), s.view(lead + (3,))


def mlp_skinning(mlp, code, pts_embed, embedding_xyz=None):
    """geom_utils.py:219-229.  With `embedding_xyz` given, pts_embed are raw positions (fused route)."""
    if mlp is None:
        return None
    return evaluate_mlp(mlp, pts_embed, embed_xyz=embedding_xyz, code=code, chunk=8 * 1024)


def _bones_arg(bones, bs, B):
    b = L.dev(bones)
    if b.dim() == 2 or b.reshape(-1, B, 10).shape[0] == 1:
        return b.reshape(B, 10), 0
    b = b.reshape(-1, B, 10)
    if b.shape[0] != bs:
        raise ValueError(f"bones: expected {bs} or 1 sets of {B} bones, got {b.shape[0]}")
    return b, 1


def _workspace(bs, B, per_ray, device):
    n = L.load().moda_warp_workspace_floats(bs, B, per_ray)
    return torch.empty((n,), device=device, dtype=torch.float32)


def _warp_autograd(bones, dq, pts, dskin, skin_aux, cyc_ref=None):
    """WarpFn on (bones, dq) in reference layout: returns (xyz_out, cyc, skin) with autograd."""
    from . import autograd as A
    bs = pts.shape[0]
    B = bones.shape[-2]
    bn = L.dev(bones).reshape(-1, B, 10)
    q = L.dev(dq).reshape(bs, B, 8)
    return A.WarpFn.apply(A.bone_prep(bn), q, L.dev(pts), None if dskin is None else L.dev(dskin), L.dev(skin_aux), cyc_ref)


def skinning(bones, pts, dskin=None, skin_aux=None):
    """geom_utils.py:280-302: bones (...,B,10), pts (bs,N,3), dskin (bs,N,B)|None -> skin (bs,N,B)."""
    bs, N, _ = pts.shape
    B = bones.shape[-2]
    if _grad(bones, pts, dskin, skin_aux):
        ident = torch.zeros((bs, B, 8), device=pts.device)
        ident[..., 0] = 1
        return _warp_autograd(bones, ident, pts, dskin, skin_aux)[2]
    b, per_ray = _bones_arg(bones, bs, B)
    p = L.dev(pts)
    d = None if dskin is None else L.dev(dskin)
    aux = L.dev(skin_aux)
    skin = torch.empty((bs, N, B), device=p.device, dtype=torch.float32)
    ws = _workspace(bs, B, per_ray, p.device)
    L.call("moda_skinning_fwd", L.ptr(b), per_ray, L.ptr(p), L.ptr(d), L.ptr(aux), bs, N, B, L.ptr(skin), L.ptr(ws),
           L.stream())
    return skin


def gauss_mlp_skinning(xyz, embedding_xyz, bones, pose_code, nerf_skin, skin_aux=None):
    """geom_utils.py:202-217."""
    dskin = mlp_skinning(nerf_skin, pose_code, xyz, embedding_xyz)
    return skinning(bones, xyz, dskin, skin_aux=skin_aux)


def dqs_blend_skinning(dq, skin, pts, _invert=0):
    """geom_utils.py:495-517: dq (bs,B,8), skin (bs,N,B), pts (bs,N,3) -> (bs,N,3).
    (Inference entry point; the differentiable route of the path is the fused WarpFn used by render_rays.)"""
    L.no_grad_only(dq, skin, pts)
    B = dq.shape[-2]
    N = pts.shape[-2]
    p = L.dev(pts).reshape(-1, N, 3)
    q = L.dev(dq).reshape(-1, B, 8)
    s = L.dev(skin).reshape(-1, N, B)
    out = torch.empty_like(p)
    L.call("moda_dqs_fwd", L.ptr(q), _invert, L.ptr(s), L.ptr(p), p.shape[0], N, B, L.ptr(out), L.stream())
    return out


def neu_dbs(bones, rts_fw, skin, xyz_in, nerf_dis=None, embedding_xyz=None, code=None, backward=True):
    """geom_utils.py:372-456 -> (xyz, bones_dfm, xyz_dis | 0).  With the residual field `nerf_dis` (off by default,
    moda.py:80): backward subtracts nerf_dis(xyz_in, code) after the blend (:416-418), forward adds it before (:420-422)."""
    B = bones.shape[-2]
    N = xyz_in.shape[-2]
    rts = rts_fw.reshape(-1, B, 8)
    pts = xyz_in.reshape(-1, N, 3)
    xyz_dis = 0
    if nerf_dis is not None:
        xyz_dis = evaluate_mlp(nerf_dis, pts, embed_xyz=embedding_xyz, code=code, chunk=pts.shape[0])   # :350-355
    if backward:
        xyz = dqs_blend_skinning(rts, skin, pts, _invert=1)
        if nerf_dis is not None:
            xyz = xyz - xyz_dis
    else:
        xyz = dqs_blend_skinning(rts, skin, pts + xyz_dis if nerf_dis is not None else pts, _invert=0)
    bones_dfm = bone_transform(bones.reshape(-1, B, 10), rts, neudbs=True)
    return xyz, bones_dfm, xyz_dis


def warp(bones, dq, pts, dskin, skin_aux, backward, want_skin=False, cyc_ref=None, dskin_bns=False, rays_per_set=1,
         pts_tf=None):
    """Fused `gauss_mlp_skinning` tail + `neu_dbs` (rendering.py:304-319 / 330-341), one kernel.
    dskin is (bs,N,B), or (bs,B,N) with dskin_bns (the layout NeRF.fused(out_tr_S=N) writes).
    rays_per_set = k > 1: frame-grouped layout, dq (bs/k, B, 8) and bones (bs/k, B, 10) | (B, 10) hold one row per k
    consecutive rays.  pts_tf (bs,N,3): the points the blended transform is applied to, when they are not `pts` (the
    points the weights are evaluated at).  Returns (xyz_out, skin|None, cyc (bs,N)|None)."""
    L.no_grad_only(bones, dq, pts, dskin, skin_aux, pts_tf)
    bs, N, _ = pts.shape
    B = bones.shape[-2]
    k = int(rays_per_set)
    if k < 1 or bs % k:
        raise ValueError(f"rays_per_set={k} does not divide {bs} rays")
    nsets = bs // k
    b, per_set = _bones_arg(bones, nsets, B)
    p = L.dev(pts)
    q = L.dev(dq).reshape(-1, B, 8)
    if q.shape[0] != nsets:
        raise ValueError(f"dq: expected {nsets} sets of {B} transforms, got {q.shape[0]}")
    d = None if dskin is None else L.dev(dskin)
    aux = L.dev(skin_aux)
    out = torch.empty_like(p)
    skin = torch.empty((bs, N, B), device=p.device) if want_skin else None
    cr = None if cyc_ref is None else L.dev(cyc_ref)
    cyc = torch.empty((bs, N), device=p.device) if cyc_ref is not None else None
    ws = _workspace(nsets, B, per_set, p.device)
    pt = None if pts_tf is None else L.dev(pts_tf)
    L.call("moda_warp_frames_fwd", L.ptr(b), per_set, L.ptr(q), k, 1 if backward else 0, L.ptr(p), L.ptr(pt), L.ptr(d),
           int(bool(dskin_bns)), L.ptr(aux), bs, N, B, L.ptr(out), L.ptr(skin), L.ptr(cr), L.ptr(cyc), L.ptr(ws), L.stream())
    return out, skin, cyc
