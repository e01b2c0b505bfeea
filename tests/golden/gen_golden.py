"""Generate golden vectors by running the MoDA reference itself (development container only).

    python tests/golden/gen_golden.py

Imports /root/reference through tests/golden/_ref_import.py, feeds it the
deterministic synthetic inputs of moda_amd/synth.py and stores the reference's
OUTPUTS (plus the random tensors it drew internally) as small .npz fixtures
next to this script.  Inputs are not stored: tests regenerate them from the
same (seed, name) keys.  Nothing here runs on the GPU box.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _ref_import import import_reference  # noqa: E402
from moda_amd import synth  # noqa: E402

rendering, nerf, geom, dq = import_reference()

# float64-truth pass (g64 below): the SAME generator functions run a second time with every floating-point input widened to
# float64 and the default dtype float64, so that the reference's own code computes in double; save() then writes
# <name>_f64.npz holding `loss` and the gradients only.  None = the ordinary fp32 run.
F64 = None          # None | "record" (fp32 run, draws recorded, nothing saved) | "replay" (float64 run on the recorded draws)
_DRAWS = []         # the tensors the reference drew in the "record" pass, in order
_F64_WRITTEN = []


def T(a):
    t = torch.from_numpy(a)
    return t.double() if (F64 == "replay" and t.is_floating_point()) else t


def save(name, **arrs):
    if F64 == "record":
        return
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    if F64 == "replay":
        out = {k: v for k, v in out.items() if k == "loss" or k.startswith("d_")}
        if not any(k.startswith("d_") for k in out):
            del _AMB[:]
            return                                  # eval-mode cases carry no gradients
        assert all(v.dtype == np.float64 for v in out.values()), {k: v.dtype for k, v in out.items()}
        # the conditioning certificate of this case: (rows of the network call, row, relative margin) per undetermined ReLU
        best = {}
        for n, r, mg in _AMB:
            best[(n, r)] = min(mg, best.get((n, r), 1.0))
        keys = sorted(best)
        out["amb_rows"] = np.asarray(keys, np.int64).reshape(-1, 2)
        out["amb_margins"] = np.asarray([best[k] for k in keys], np.float64)
        del _AMB[:]
        name += "_f64"
        _F64_WRITTEN.append(name)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


AMB_TOL = 2e-6      # a ReLU pre-activation within this (relative to its layer's largest) of zero is undetermined at fp32 accuracy
_AMB = []           # float64 pass: (rows of the call, row index, relative margin) of every such pre-activation


def _watch_relus(m):
    """float64 pass: a forward hook on every Linear that feeds a ReLU (nerf.py:100-137: Sequential(Linear, ReLU(True))) notes the
    rows (samples) holding a pre-activation with |z| < AMB_TOL max|z|: there two correct fp32 evaluations may disagree on the
    ReLU's state, and that sample's gradient is not determined at fp32 accuracy.  The fixture carries these rows, so the GPU tests
    know which rays can be compared at the tight bar (conditioning certificate, computed by the reference itself in float64)."""
    for seq in m.modules():
        if isinstance(seq, torch.nn.Sequential) and len(seq) >= 2 and isinstance(seq[0], torch.nn.Linear) and isinstance(seq[1], torch.nn.ReLU):
            def hook(mod, inp, out):
                z = out.detach().reshape(-1, out.shape[-1]).abs()
                rel = z / z.max()
                rows = (rel < AMB_TOL).any(-1).nonzero().reshape(-1)
                for r in rows.tolist():
                    _AMB.append((z.shape[0], r, float(rel[r].min())))
            seq[0].register_forward_hook(hook)


def ref_nerf(p, **kw):
    m = nerf.NeRF(**kw)
    if F64 == "replay":
        m = m.double()
        _watch_relus(m)
    m.load_state_dict({k: T(v) for k, v in p.items()})
    return m.eval()


class RecordRandom:
    """Record (and optionally replace) the tensors rendering.py draws from torch.rand / rand_like / randn."""

    def __init__(self):
        self.log = []

    def __enter__(self):
        self._orig = (torch.rand, torch.rand_like, torch.randn, torch.randn_like)
        rec = self

        def draw(i, kind, a, k):
            t = rec._orig[i](*a, **k)
            rec.log.append((kind, t.clone()))
            return t

        def randn_like(*a, **k):
            return draw(3, "randn_like", a, k)

        def rand(*a, **k):
            return draw(0, "rand", a, k)

        def rand_like(*a, **k):
            return draw(1, "rand_like", a, k)

        def randn(*a, **k):
            return draw(2, "randn", a, k)

        torch.rand, torch.rand_like, torch.randn, torch.randn_like = rand, rand_like, randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.rand, torch.rand_like, torch.randn, torch.randn_like = self._orig


# --------------------------------------------------------------------------- G1 dual_quat
def g1():
    a = T(synth.normal(1, "g1/a", (37, 8)))
    b = T(synth.normal(1, "g1/b", (37, 8)))
    save("g1_dual_quat",
         q_mul=dq.q_mul(a[:, :4], b[:, :4]), dq_mul=dq.dq_mul(a, b), dq_normalize=dq.dq_normalize(a),
         dq_inverse=dq.dq_inverse(a), dq_qconj=dq.dq_quaternion_conjugate(a),
         dq_cconj=dq.dq_combined_conjugate(a), q_normalize=dq.q_normalize(a[:, :4]),
         dq_mul_nd=dq.dq_mul(a.view(1, 37, 8), b.view(1, 37, 8)))


# --------------------------------------------------------------------------- G2 Embedding
def g2():
    x = T(synth.normal(2, "g2/x", (5, 7, 3)))
    out = {}
    for alpha in (6.5, 10.0):
        out[f"xyz_a{alpha}"] = nerf.Embedding(3, 10, alpha=alpha)(x)
        out[f"dir_a{alpha}"] = nerf.Embedding(3, 4, alpha=alpha)(x)
    out["xyz_default"] = nerf.Embedding(3, 10)(x)
    save("g2_embedding", **out)


# --------------------------------------------------------------------------- G3 NeRF.forward
NERF_SHAPES = {
    "coarse": dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=27 + 64, out_channels=3, raw_feat=False),
    "skin": dict(D=5, W=64, in_channels_xyz=63 + 128, in_channels_dir=0, out_channels=25, raw_feat=True,
                 in_channels_code=128),
    "feat": dict(D=5, W=128, in_channels_xyz=63, in_channels_dir=0, out_channels=16, raw_feat=True),
    "vis": dict(D=5, W=64, in_channels_xyz=63, in_channels_dir=0, out_channels=1, raw_feat=True),
}


def g3():
    out = {}
    M = 257
    for name, kw in NERF_SHAPES.items():
        pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
        p = synth.nerf_params(3, "g3/" + name, **pk)
        m = ref_nerf(p, **kw)
        x = T(synth.normal(3, "g3/x/" + name, (M, kw["in_channels_xyz"] + kw["in_channels_dir"])))
        with torch.no_grad():
            out[name] = m(x)
            out[name + "_sigma"] = m(x[:, :kw["in_channels_xyz"]], sigma_only=True)
    save("g3_nerf", **out)


# --------------------------------------------------------------------------- G4 skinning / DQS
def g4():
    out = {}
    for B in (25, 36):
        N, S = 12, 9
        bones = synth.make_models(4, B=B, with_skin=False, perturb_bones=True)["bones_rst"]
        rts = synth.frame_dual_quats(4, f"g4/rts{B}", N, B)
        xyz = np.float32(0.2) * synth.normal(4, f"g4/xyz{B}", (N, S, 3))
        dskin = synth.normal(4, f"g4/dskin{B}", (N, S, B))
        skin_aux = T(np.asarray([0.3, 10], np.float32))
        with torch.no_grad():
            bd = geom.bone_transform(T(bones), T(rts), True, is_vec=True)
            out[f"bone_transform_{B}"] = bd
            out[f"skin_ray_dskin_{B}"] = geom.skinning(bd, T(xyz), T(dskin), skin_aux)
            out[f"skin_ray_{B}"] = geom.skinning(bd, T(xyz), None, skin_aux)
            out[f"skin_rest_dskin_{B}"] = geom.skinning(T(bones), T(xyz), T(dskin), skin_aux)
            skin = out[f"skin_ray_dskin_{B}"]
            out[f"dqs_{B}"] = geom.dqs_blend_skinning(T(rts).view(N, B, 8), skin, T(xyz))
            out[f"neu_dbs_bw_{B}"] = geom.neu_dbs(T(bones), T(rts), skin, T(xyz), backward=True)[0]
            out[f"neu_dbs_fw_{B}"] = geom.neu_dbs(T(bones), T(rts), skin, T(xyz), backward=False)[0]
    save("g4_skinning", **out)


# --------------------------------------------------------------------------- shared scene builders
def make_opts(**kw):
    o = dict(dist_corresp=False, lbs=False, neudbs=True, symm_shape=False, scale_rgb=1.3, rgb_filter=False,
             use_corresp=False, use_corr=False, use_ot=False, s3im_loss=False)
    o.update(kw)
    return types.SimpleNamespace(**o)


def ref_scene(seed, B, with_skin=True, with_feat=False, with_vis=False, alpha=10.0, perturb_bones=False, with_dis=False):
    mp = synth.make_models(seed, B=B, with_skin=with_skin, with_feat=with_feat, with_vis=with_vis,
                           perturb_bones=perturb_bones, with_dis=with_dis)
    models = {"coarse": ref_nerf(mp["coarse"], in_channels_xyz=63, in_channels_dir=27 + 64, init_beta=0.1)}
    if B > 0:
        models["bones"] = torch.nn.Parameter(T(mp["bones_rst"]))
        models["bones_rst"] = T(mp["bones_rst"])
        models["skin_aux"] = T(mp["skin_aux"])
        rpc = torch.nn.Embedding(1, 128)
        if with_skin:
            models["nerf_skin"] = ref_nerf(mp["nerf_skin"], **{**NERF_SHAPES["skin"], "out_channels": B})
            rpc.weight.data = T(mp["rest_pose_code"])
        models["rest_pose_code"] = rpc
    if with_dis:
        models["nerf_dis"] = ref_nerf(mp["nerf_dis"], D=5, W=128, in_channels_xyz=63 + 128, in_channels_dir=0,
                                      out_channels=3, raw_feat=True, in_channels_code=128)
    if with_feat:
        models["nerf_feat"] = ref_nerf(mp["nerf_feat"], **NERF_SHAPES["feat"])
    if with_vis:
        models["nerf_vis"] = ref_nerf(mp["nerf_vis"], **NERF_SHAPES["vis"])
    emb = {"xyz": nerf.Embedding(3, 10, alpha=alpha), "dir": nerf.Embedding(3, 4, alpha=alpha)}
    return models, emb


E2E_KEYS = ("img_coarse", "depth_rnd", "sil_coarse", "xyz_camera_vis", "xyz_canonical_vis", "frame_cyc_dis",
            "vis_pred", "dis_reg", "dis_reg_forward")


def run_ref(seed, N, S, B, rays_per_frame=16, opts=None, **kw):
    scene_kw = {k: kw.pop(k) for k in ("with_skin", "with_feat", "with_vis", "alpha", "perturb_bones", "with_dis") if k in kw}
    models, emb = ref_scene(seed, B, **scene_kw)
    rays = {k: T(v) for k, v in synth.make_rays(seed, N, B, rays_per_frame=rays_per_frame).items()}
    opts = opts or make_opts()
    with RecordRandom() as rec, torch.no_grad():
        res = rendering.render_rays(models, emb, rays, N_samples=S, chunk=1024 * 32, img_size=512, opts=opts, **kw)
    out = {k: res[k] for k in E2E_KEYS if k in res}
    for i, (kind, t) in enumerate(rec.log):
        out[f"rng{i}_{kind}"] = t
    return out


# --------------------------------------------------------------------------- G5 compositing
def g5():
    N, S = 9, 12
    models, emb = ref_scene(5, 0)
    rays = synth.make_rays(5, N, 0)
    z = np.sort(np.float32(0.1) + np.float32(0.4) * synth.uniform(5, "g5/z", (N, S)), -1).astype(np.float32)
    xyz = rays["rays_o"][:, None] + rays["rays_d"][:, None] * z[:, :, None]
    xyz[3] += 10.0  # a ray far outside the object: near-zero density everywhere but the last bin
    d_emb = emb["dir"](T(rays["rays_d"]))
    vis_pred = T(synth.uniform(5, "g5/vis", (N, S)))
    torch.manual_seed(5)
    with RecordRandom() as rec, torch.no_grad():
        o1 = rendering.inference(models, emb["xyz"], T(xyz), T(rays["rays_d"]), d_emb, T(z), N, S, 4096, 0.5,
                                 env_code=T(rays["env_code"]))
        o2 = rendering.inference(models, emb["xyz"], T(xyz), T(rays["rays_d"]), d_emb, T(z), N, S, 4096, 0.0,
                                 env_code=T(rays["env_code"]), clip_bound=[0.12, 0.12, 0.25], vis_pred=vis_pred)
    names = ("rgb", "feat", "depth", "weights", "vis", "sil")
    out = {"z": z, "xyz": xyz}
    for tag, o in (("noise", o1), ("mask", o2)):
        for n, v in zip(names, o):
            out[f"{tag}_{n}"] = v
    out["noise_randn"] = rec.log[0][1]
    save("g5_composite", **out)


# --------------------------------------------------------------------------- G6 sample_pdf
def g6():
    N, S = 11, 14
    bins = np.sort(synth.uniform(6, "g6/bins", (N, S + 1)), -1).astype(np.float32)
    w = synth.uniform(6, "g6/w", (N, S)).astype(np.float32)
    w[2] = 0  # all-zero weights: uniform pdf after the eps
    w[4, 3:9] = 0  # zero-weight bins: denom < eps branch
    u = synth.uniform(6, "g6/u", (N, 20))
    orig = torch.rand
    torch.rand = lambda *a, **k: T(u)
    try:
        rnd = rendering.sample_pdf(T(bins), T(w), 20, det=False)
    finally:
        torch.rand = orig
    det = rendering.sample_pdf(T(bins), T(w), 20, det=True)
    save("g6_sample_pdf", det=det, rnd=rnd)


# --------------------------------------------------------------------------- G7 end to end (small)
def g7():
    N, S, B = 64, 16, 25
    torch.manual_seed(7)
    cases = {
        "nobones": dict(B=0),
        "bones_noskin": dict(B=B, with_skin=False),
        "bones_skin": dict(B=B),
        "bones36_skin": dict(B=36, perturb_bones=True),
        "alpha65": dict(B=B, alpha=6.5),
        "perturb": dict(B=B, perturb=1.0, noise_std=0.3),
        "symm": dict(B=B, opts=make_opts(symm_shape=True)),
        "fine": dict(B=B, use_fine=True, S=32),
        "fine_perturb_symm": dict(B=B, use_fine=True, S=32, perturb=1.0, noise_std=0.2,
                                  opts=make_opts(symm_shape=True)),
        "feat": dict(B=B, with_feat=True),
        "render_vis": dict(B=B, with_vis=True, render_vis=True, obj_bound=np.asarray([0.15, 0.15, 0.15])),
        "disp": dict(B=B, use_disp=True),
        "rgb_filter": dict(B=B, opts=make_opts(rgb_filter=True)),
        "dis": dict(B=B, with_dis=True),
        "dis_fine": dict(B=B, with_dis=True, use_fine=True, S=32),
    }
    only = os.environ.get("G7_ONLY")
    for name, kw in cases.items():
        if only and name not in only.split(","):
            continue
        kw = dict(kw)
        B_ = kw.pop("B")
        S_ = kw.pop("S", S)
        kw.setdefault("noise_std", 0.0)
        save("g7_" + name, **run_ref(7, N, S_, B_, **kw))


# --------------------------------------------------------------------------- G8 cfg1 checksum
def g8():
    """BASELINE config 1: 4096 rays x 64 samples, 25 bones, eval / no_grad, perturb=0, noise_std=0."""
    out = run_ref(0, 4096, 64, 25, rays_per_frame=256, noise_std=0.0)
    idx = np.arange(0, 4096, 256)
    small = {}
    for k, v in out.items():
        if k.startswith("rng"):
            continue
        a = v.numpy()
        small[k + "_mean"] = a.astype(np.float64).mean()
        small[k + "_absmax"] = np.abs(a).max()
        small[k + "_rays"] = a[idx]
    small["ray_index"] = idx
    save("g8_cfg1", **small)




# --------------------------------------------------------------------------- G9 gradients (training mode)
GRAD_LEAVES = ("rays_o", "rays_d", "bone_rts", "time_embedded", "env_code")


def g9():
    """Gradients of a fixed scalar of the rendered outputs w.r.t. parameters and ray inputs (reference autograd).
    L = sum_k <c_k, out_k> over img_coarse, depth_rnd, sil_coarse, frame_cyc_dis with fixed pseudo-random c_k."""
    N, S = 48, 12
    for name, B, with_skin in (("nobones", 0, False), ("bones_noskin", 25, False), ("bones_skin", 25, True)):
        models, emb = ref_scene(9, B, with_skin=with_skin, perturb_bones=True)
        for m in models.values():
            if isinstance(m, torch.nn.Module):
                m.train()
        if B > 0:
            models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
            models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
        rays = {k: T(v) for k, v in synth.make_rays(9, N, B, rays_per_frame=8).items()}
        for k in GRAD_LEAVES:
            if k in rays:
                rays[k].requires_grad_(True)
        torch.manual_seed(9)
        with RecordRandom() as rec:
            res = rendering.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, chunk=1024 * 32, img_size=512,
                                        opts=make_opts())
        keys = [k for k in ("img_coarse", "depth_rnd", "sil_coarse", "frame_cyc_dis") if k in res]
        loss = 0
        for k in keys:
            c = T(synth.normal(9, "g9/c/" + k, tuple(res[k].shape)))
            loss = loss + (c * res[k]).sum()
        loss.backward()
        out = {"loss": loss.detach()}

        def put(key, g):
            """Small gradients whole; 256x256-class ones as a 16x16 corner + norm + sum (fixtures stay small)."""
            if g is None:
                return
            if g.numel() <= 20000:
                out[key] = g
            else:
                out[key + "__corner"] = g[:16, :16].clone()
                out[key + "__norm"] = g.double().norm()
                out[key + "__sum"] = g.double().sum()

        for k in GRAD_LEAVES:
            if k in rays:
                put("d_" + k, rays[k].grad)
        for mname in ("coarse", "nerf_skin"):
            if mname in models:
                for pn, p in models[mname].named_parameters():
                    put(f"d_{mname}.{pn}", p.grad)
        if B > 0:
            put("d_bones_rst", models["bones_rst"].grad)
            put("d_skin_aux", models["skin_aux"].grad)
            if with_skin:
                put("d_rest_pose_code", models["rest_pose_code"].weight.grad)
        save("g9_grad_" + name, **out)


# --------------------------------------------------------------------------- G10 correspondence / loss block
G10_KEYS = ("img_coarse", "sil_coarse", "flo_coarse", "flo_valid", "fdp_coarse", "fdp_valid", "img_loss_samp",
            "sil_loss_samp", "flo_loss_samp", "sil_at_samp_flo", "frame_cyc_dis")
G10_LOSS = ("flo_coarse", "fdp_coarse", "img_loss_samp", "sil_loss_samp", "flo_loss_samp", "frame_cyc_dis")


def g10():
    """inference_deform with the paired-frame keys (dist_corresp) and the observed-signal keys: flow rendering and the
    img / sil / flo loss terms (rendering.py:345-360, 439-499, 518-571); eval and train mode, plus gradients."""
    N, S, B = 48, 12, 25
    for mode in ("eval", "train"):
        models, emb = ref_scene(10, B, with_skin=True, perturb_bones=True)
        if mode == "train":
            models["coarse"].train()
            models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
            models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
        rays = {k: T(v) for k, v in synth.make_rays(10, N, B, rays_per_frame=8).items()}
        rays.update({k: T(v) for k, v in synth.make_corresp_rays(10, N, B, rays_per_frame=8).items()})
        leaves = ("rays_o", "rays_d", "bone_rts", "bone_rts_target", "rtk_vec_target", "time_embedded")
        if mode == "train":
            for k in leaves:
                rays[k].requires_grad_(True)
        ctx = torch.enable_grad() if mode == "train" else torch.no_grad()
        with ctx:
            res = rendering.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, chunk=1024 * 32, img_size=512,
                                        opts=make_opts(dist_corresp=True, use_corresp=True))
        out = {k: res[k].detach().float() for k in G10_KEYS if k in res}
        if mode == "train":
            loss = 0
            for k in G10_LOSS:
                loss = loss + (T(synth.normal(10, "g10/c/" + k, tuple(res[k].shape))) * res[k]).sum()
            loss.backward()
            out["loss"] = loss.detach()
            for k in leaves:
                out["d_" + k] = rays[k].grad
            out["d_bones_rst"] = models["bones_rst"].grad
            out["d_coarse.sigma.weight"] = models["coarse"].sigma.weight.grad
            out["d_coarse.xyz_encoding_1.0.weight"] = models["coarse"].xyz_encoding_1[0].weight.grad
            out["d_nerf_skin.rgb.0.weight"] = models["nerf_skin"].rgb[0].weight.grad
        save("g10_corresp_" + mode, **out)


# --------------------------------------------------------------------------- G11 feature / visibility heads
G11_KEYS = ("img_coarse", "sil_coarse", "pts_pred", "pts_exp", "feat_err", "proj_err", "vis_loss", "frnd_loss_samp",
            "flo_coarse", "img_loss_samp", "sil_loss_samp", "flo_loss_samp", "frame_cyc_dis")
G11_LOSS = ("pts_pred", "feat_err", "proj_err", "vis_loss", "frnd_loss_samp", "flo_coarse", "img_loss_samp",
            "sil_loss_samp", "frame_cyc_dis")
G11_BOUND = np.asarray([0.2, 0.2, 0.2], np.float32)


def g11():
    """The full training configuration of inference_deform (moda.py defaults: use_embed, use_proj, use_corresp,
    dist_corresp, nerf_vis, use_ot): CSE feature matching (Sinkhorn and softmax forms) + keypoint reprojection +
    visibility loss + rendered-feature loss (rendering.py:417-437, 475-477, 573-578), outputs and gradients."""
    N, S, B = 48, 12, 25
    for mode, use_ot in (("eval_ot", True), ("train_ot", True), ("train_softmax", False)):
        train = mode.startswith("train")
        models, emb = ref_scene(11, B, with_skin=True, with_feat=True, with_vis=True, perturb_bones=True)
        if train:
            for m in models.values():
                if isinstance(m, torch.nn.Module):
                    m.train()
            models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
            models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
        rays = {k: T(v) for k, v in synth.make_rays(11, N, B, rays_per_frame=8).items()}
        rays.update({k: T(v) for k, v in synth.make_corresp_rays(11, N, B, rays_per_frame=8).items()})
        rays.update({k: T(v) for k, v in synth.make_feat_rays(11, N, rays_per_frame=8).items()})
        leaves = ("rays_o", "rays_d", "bone_rts", "rtk_vec", "time_embedded")
        if train:
            for k in leaves:
                rays[k].requires_grad_(True)
        torch.manual_seed(11)
        with RecordRandom() as rec, (torch.enable_grad() if train else torch.no_grad()):
            res = rendering.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, chunk=1024 * 32, img_size=512,
                                        obj_bound=G11_BOUND,
                                        opts=make_opts(dist_corresp=True, use_corresp=True, use_ot=use_ot))
        out = {k: res[k].detach().float() for k in G11_KEYS if k in res}
        for i, (kind, t) in enumerate(rec.log):
            if kind != "randn":                     # the (N,S) density noise is unused at noise_std = 0
                out[f"rng_{kind}"] = t
        if train:
            loss = 0
            for k in G11_LOSS:
                c = T(synth.normal(11, "g11/c/" + k, tuple(res[k].shape) or (1,))).reshape(res[k].shape)
                loss = loss + (c * res[k]).sum()
            loss.backward()
            out["loss"] = loss.detach()
            for k in leaves:
                out["d_" + k] = rays[k].grad
            out["d_bones_rst"] = models["bones_rst"].grad
            out["d_nerf_feat.beta"] = models["nerf_feat"].beta.grad if models["nerf_feat"].beta.grad is not None \
                else torch.zeros(1)
            for mn, pn in (("nerf_feat", "rgb.0.weight"), ("nerf_feat", "xyz_encoding_1.0.weight"),
                           ("nerf_vis", "rgb.0.weight"), ("nerf_vis", "xyz_encoding_1.0.weight"),
                           ("coarse", "sigma.weight"), ("nerf_skin", "rgb.0.weight")):
                out[f"d_{mn}.{pn}"] = dict(models[mn].named_parameters())[pn].grad
        save("g11_heads_" + mode, **out)


# --------------------------------------------------------------------------- G12 per-frame feeders
G12 = dict(F=6, ns=11, B=5, code=32, vid_offset=[0, 40, 100], n_freq=6)


def g12():
    """The feeders immediately before the path: raycast (geom_utils.py:746-794), FrameCode
    (nerf.py:346-380), DQ_RTHead (nerf.py:239-279), correct_rest_pose (geom_utils.py:953-972): outputs and gradients."""
    F_, ns, B, C = G12["F"], G12["ns"], G12["B"], G12["code"]
    out = {}
    cam = {k: T(v).requires_grad_(k != "near_far") for k, v in synth.make_cameras(12, F_).items()}
    xys = T(synth.uniform(12, "g12/xys", (F_, ns, 2)) * np.float32(512))
    for tag, nf in (("nf", cam["near_far"]), ("auto", None)):
        rays = geom.raycast(xys, cam["Rmat"], cam["Tmat"], cam["Kinv"], nf)
        for k in ("rays_o", "rays_d", "near", "far", "rtk_vec"):
            out[f"raycast_{tag}_{k}"] = rays[k].detach()
        if tag == "nf":
            loss = (T(synth.normal(12, "g12/c/d", (F_, ns, 3))) * rays["rays_d"]).sum() \
                + (T(synth.normal(12, "g12/c/o", (F_, ns, 3))) * rays["rays_o"]).sum()
            loss.backward()
            for k in ("Rmat", "Tmat", "Kinv"):
                out["raycast_d_" + k] = cam[k].grad.clone()
    # FrameCode
    fc = nerf.FrameCode(G12["n_freq"], C, np.asarray(G12["vid_offset"]))
    w, b = synth.linear_init(12, "g12/fc", C, fc.basis_mlp.in_features)
    fc.basis_mlp.weight.data, fc.basis_mlp.bias.data = T(w), T(b)
    fid = torch.tensor([0, 3, 39, 40, 41, 77, 99, 12])
    code = fc(fid)
    (T(synth.normal(12, "g12/c/code", tuple(code.shape))) * code).sum().backward()
    out["framecode"] = code.detach()
    out["framecode_d_weight"] = fc.basis_mlp.weight.grad.clone()
    # DQ_RTHead
    kw = dict(D=8, W=64, in_channels_xyz=C, in_channels_dir=0, out_channels=7 * B, raw_feat=True)
    head = nerf.DQ_RTHead(use_quat=True, **kw)
    pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
    head.load_state_dict({k: T(v) for k, v in synth.nerf_params(12, "g12/head", **pk).items()})
    x = T(synth.normal(12, "g12/x", (8, C))).requires_grad_(True)
    dq = head(x)
    (T(synth.normal(12, "g12/c/dq", tuple(dq.shape))) * dq).sum().backward()
    out["rthead"] = dq.detach()
    out["rthead_d_x"] = x.grad.clone()
    out["rthead_d_rgb"] = head.rgb[0].weight.grad.clone()
    out["rthead_d_l5"] = head.xyz_encoding_5[0].weight.grad.clone()
    # correct_rest_pose
    fw = T(synth.frame_dual_quats(12, "g12/fw", 7, B)).requires_grad_(True)
    rst = T(synth.frame_dual_quats(12, "g12/rst", 1, B)).requires_grad_(True)
    o = types.SimpleNamespace(num_bones=B)
    delta = geom.correct_rest_pose(o, fw[:, None], rst, True)
    (T(synth.normal(12, "g12/c/delta", tuple(delta.shape))) * delta).sum().backward()
    out["rest_delta"] = delta.detach()
    out["rest_d_fw"], out["rest_d_rst"] = fw.grad.clone(), rst.grad.clone()
    save("g12_feeders", **out)


# --------------------------------------------------------------------------- G13 canonical-grid / mesh-extraction queries
G13 = dict(B=25, P=200, grid=6, embedid=3, vid_offset=[0, 50], code=128)


def g13_params():
    """Deterministic parameters of the per-frame modules a MoDA model holds for warp_bw / warp_fw (moda.py:282-330)."""
    B, C = G13["B"], G13["code"]
    hk = dict(D=8, W=256, in_channels_xyz=C, in_channels_dir=0, out_channels=7 * B)
    head = synth.nerf_params(13, "g13/head", **hk)
    head["rgb.0.weight"] = head["rgb.0.weight"] * np.float32(0.05)
    head["rgb.0.bias"] = np.tile(np.asarray([0, 0, 0, 1, 0, 0, 0], np.float32), B) \
        + np.float32(0.1) * synth.normal(13, "g13/head/b", (7 * B,))
    fw, fb = synth.linear_init(13, "g13/pose", C, 2 * (1 + 2 * 6))    # FrameCode(num_freq=6): 13 channels x 1 video
    return head, fw, fb


def g13():
    """warp_bw / warp_fw (geom_utils.py:974-1073) for a mock model, and the volume queries of extract_mesh
    (train_utils.py:1378-1422: nerf_coarse(sigma_only) and sigmoid(nerf_vis) on a canonical grid)."""
    B, P, C = G13["B"], G13["P"], G13["code"]
    head_p, fw, fb = g13_params()
    mp = synth.make_models(13, B=B, with_skin=True, with_vis=True, perturb_bones=True, with_dis=True)
    model = types.SimpleNamespace(device="cpu")
    model.embedding_xyz = nerf.Embedding(3, 10, alpha=10.0)
    model.pose_code = nerf.FrameCode(6, C, np.asarray(G13["vid_offset"]))
    model.pose_code.basis_mlp.weight.data, model.pose_code.basis_mlp.bias.data = T(fw[:, :model.pose_code.basis_mlp.in_features]), T(fb)
    head = nerf.DQ_RTHead(use_quat=True, in_channels_xyz=C, in_channels_dir=0, out_channels=7 * B, raw_feat=True)
    head.load_state_dict({k: T(v) for k, v in head_p.items()})
    model.nerf_body_rts = torch.nn.Sequential(model.pose_code, head)
    model.bones = T(mp["bones_rst"])
    model.rest_pose_code = torch.nn.Embedding(1, C)
    model.rest_pose_code.weight.data = T(mp["rest_pose_code"])
    model.nerf_skin = ref_nerf(mp["nerf_skin"], **{**NERF_SHAPES["skin"], "out_channels": B})
    model.skin_aux = T(mp["skin_aux"])
    model.opts = types.SimpleNamespace(num_bones=B)
    opts = types.SimpleNamespace(flowbw=False, lbs=False, neudbs=True, nerf_skin=True, nerf_dis=False, num_bones=B)
    pts = np.float32(0.15) * synth.normal(13, "g13/pts", (P, 3))
    out = {}
    with torch.no_grad():
        bw, d1 = geom.warp_bw(opts, model, {}, T(pts).clone(), G13["embedid"])
        fwp, d2 = geom.warp_fw(opts, model, {}, pts.copy(), G13["embedid"])
        out["warp_bw"], out["warp_bw_bones"] = bw, d1["bones"]
        out["warp_fw"], out["warp_fw_bones"] = T(fwp), d2["bones"]
        # the same warps with the residual displacement field switched on (geom_utils.py:1010-1022, 1060-1069)
        model.nerf_dis = ref_nerf(mp["nerf_dis"], D=5, W=128, in_channels_xyz=63 + 128, in_channels_dir=0, out_channels=3,
                                  raw_feat=True, in_channels_code=128)
        opts_d = types.SimpleNamespace(**{**vars(opts), "nerf_dis": True})
        out["warp_bw_dis"] = geom.warp_bw(opts_d, model, {}, T(pts).clone(), G13["embedid"])[0]
        out["warp_fw_dis"] = T(geom.warp_fw(opts_d, model, {}, pts.copy(), G13["embedid"])[0])
        # volume queries
        gs, bound = G13["grid"], np.asarray([0.2, 0.15, 0.25], np.float32)
        ax = [np.linspace(-bound[c], bound[c], gs).astype(np.float32) for c in range(3)]
        q = np.stack(np.meshgrid(ax[1], ax[0], ax[2]), -1).reshape(-1, 3)[:, [1, 0, 2]]       # (x, y, z), train_utils.py:1381-1389
        coarse = ref_nerf(mp["coarse"], in_channels_xyz=63, in_channels_dir=27 + 64, init_beta=0.1)
        vis = ref_nerf(mp["nerf_vis"], **NERF_SHAPES["vis"])
        e = model.embedding_xyz(T(q))
        out["vol_sigma"] = coarse(e, sigma_only=True).view(gs, gs, gs)
        out["vol_vis"] = vis(e)[..., 0].sigmoid().view(gs, gs, gs)
    save("g13_grid", **out)


# --------------------------------------------------------------------------- G14 eikonal regulariser
G14 = dict(shape=(7, 9, 3), bound=[0.2, 0.15, 0.25], scale=0.12)
G14_GRADS = ("xyz_encoding_1.0.weight", "xyz_encoding_1.0.bias", "xyz_encoding_5.0.weight", "xyz_encoding_8.0.weight",
             "xyz_encoding_8.0.bias", "sigma.weight")


def g14():
    """eikonal_loss (loss_utils.py:73-104) on nerf_coarse: the analytic form (nerf_gradient :15-47, whose parameter
    gradients need the double backward) and the finite-difference form (compute_gradients_sdf :48-71)."""
    import importlib
    lu = importlib.import_module("nnutils.loss_utils")
    mp = synth.make_models(14, B=0)
    emb = nerf.Embedding(3, 10, alpha=10.0)
    pts = np.float32(G14["scale"]) * synth.normal(14, "g14/pts", G14["shape"])
    out = {}
    for tag, ppr in (("ana", False), ("fd", True)):
        coarse = ref_nerf(mp["coarse"], in_channels_xyz=63, in_channels_dir=27 + 64, init_beta=0.1)
        loss = lu.eikonal_loss(coarse, emb, T(pts).clone(), G14["bound"], ppr)
        loss.backward()
        out[tag + "_loss"] = loss.detach()
        sd = dict(coarse.named_parameters())
        for k in G14_GRADS:
            out[f"{tag}_d_{k}"] = sd[k].grad.clone()
    g, sig = lu.nerf_gradient(ref_nerf(mp["coarse"], in_channels_xyz=63, in_channels_dir=27 + 64, init_beta=0.1), emb,
                              T(pts).clone().view(1, -1, 3), sigma_only=True)
    out["grad"], out["sigmas"] = g.detach(), sig.detach()
    save("g14_eikonal", **out)


# --------------------------------------------------------------------------- G15 residual displacement field (nerf_dis)
G15_KEYS = ("img_coarse", "sil_coarse", "depth_rnd", "xyz_canonical_vis", "frame_cyc_dis", "dis_reg", "dis_reg_forward",
            "flo_coarse", "flo_valid", "fdp_coarse", "fdp_valid")
G15_LOSS = ("img_coarse", "frame_cyc_dis", "dis_reg", "dis_reg_forward", "flo_coarse", "fdp_coarse")
G15_LEAVES = ("rays_o", "rays_d", "bone_rts", "bone_rts_target", "time_embedded")


def g15():
    """inference_deform with the residual displacement field nerf_dis (moda.py:80,334-340; geom_utils.py:350-355, 416-422;
    rendering.py:307-322, 342-360) and the paired-frame warps that reuse it: outputs (eval) and gradients (train)."""
    N, S, B = 48, 12, 25
    for mode in ("eval", "train"):
        models, emb = ref_scene(15, B, with_skin=True, perturb_bones=True, with_dis=True)
        if mode == "train":
            for m in models.values():
                if isinstance(m, torch.nn.Module):
                    m.train()
            models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
        rays = {k: T(v) for k, v in synth.make_rays(15, N, B, rays_per_frame=8).items()}
        rays.update({k: T(v) for k, v in synth.make_corresp_rays(15, N, B, rays_per_frame=8).items()})
        if mode == "train":
            for k in G15_LEAVES:
                rays[k].requires_grad_(True)
        with (torch.enable_grad() if mode == "train" else torch.no_grad()):
            res = rendering.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, chunk=1024 * 32, img_size=512,
                                        opts=make_opts(dist_corresp=True))
        out = {k: res[k].detach().float() for k in G15_KEYS if k in res}
        if mode == "train":
            loss = 0
            for k in G15_LOSS:
                loss = loss + (T(synth.normal(15, "g15/c/" + k, tuple(res[k].shape))) * res[k]).sum()
            loss.backward()
            out["loss"] = loss.detach()
            for k in G15_LEAVES:
                out["d_" + k] = rays[k].grad
            out["d_bones_rst"] = models["bones_rst"].grad
            for mn, pn in (("nerf_dis", "rgb.0.weight"), ("nerf_dis", "rgb.0.bias"), ("nerf_dis", "xyz_encoding_1.0.weight"),
                           ("nerf_dis", "xyz_encoding_5.0.weight"), ("nerf_skin", "rgb.0.weight"), ("coarse", "sigma.weight")):
                out[f"d_{mn}.{pn}"] = dict(models[mn].named_parameters())[pn].grad
            out["d_rest_pose_code"] = models["rest_pose_code"].weight.grad
        save("g15_dis_" + mode, **out)


# --------------------------------------------------------------------------- G16 rgb_filter gradients
G16_LEAVES = ("rays_o", "rays_d", "bone_rts", "env_code")
G16_PARAMS = ("sigma.weight", "sigma.bias", "rgb.0.weight", "xyz_encoding_8.0.weight", "beta")


def g16():
    """opts.rgb_filter (rendering.py:171, 225-230): colour weighted by scale_rgb * sigmoid(-10 sigma_raw), last sample
    excluded -- outputs and gradients (the extra path into sigma through the semantic weight)."""
    N, S, B = 48, 12, 25
    models, emb = ref_scene(16, B, with_skin=True, perturb_bones=True)
    for m in models.values():
        if isinstance(m, torch.nn.Module):
            m.train()
    rays = {k: T(v) for k, v in synth.make_rays(16, N, B, rays_per_frame=8).items()}
    for k in G16_LEAVES:
        rays[k].requires_grad_(True)
    res = rendering.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, chunk=1024 * 32, img_size=512,
                                opts=make_opts(rgb_filter=True))
    out = {k: res[k].detach() for k in ("img_coarse", "depth_rnd", "sil_coarse")}
    loss = 0
    for k in ("img_coarse", "depth_rnd", "sil_coarse"):
        loss = loss + (T(synth.normal(16, "g16/c/" + k, tuple(res[k].shape))) * res[k]).sum()
    loss.backward()
    out["loss"] = loss.detach()
    for k in G16_LEAVES:
        out["d_" + k] = rays[k].grad
    sd = dict(models["coarse"].named_parameters())
    for k in G16_PARAMS:
        out["d_coarse." + k] = sd[k].grad
    save("g16_rgb_filter", **out)


# --------------------------------------------------------------------------- G17 back-correspondence term (use_corr)
G17_LOSS = ("pts_pred", "feat_err", "corr_err", "proj_err")


def g17():
    """opts.use_corr (moda.py:157, off by default): corr_err = |P P^T - I|_2 per pixel on the matching probabilities
    (loss_utils.py:386-391) for the Sinkhorn and the softmax forms, with the gradients it sends back through them."""
    N, S, B = 48, 12, 25
    for mode, use_ot in (("ot", True), ("softmax", False)):
        models, emb = ref_scene(17, B, with_skin=True, with_feat=True, perturb_bones=True)
        for m in models.values():
            if isinstance(m, torch.nn.Module):
                m.train()
        rays = {k: T(v) for k, v in synth.make_rays(17, N, B, rays_per_frame=8).items()}
        rays.update({k: T(v) for k, v in synth.make_corresp_rays(17, N, B, rays_per_frame=8).items()})
        rays.update({k: T(v) for k, v in synth.make_feat_rays(17, N, rays_per_frame=8).items()})
        torch.manual_seed(17)
        with RecordRandom() as rec:
            res = rendering.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, chunk=1024 * 32, img_size=512,
                                        obj_bound=G11_BOUND, opts=make_opts(dist_corresp=True, use_corresp=True, use_ot=use_ot,
                                                                            use_corr=True))
        out = {k: res[k].detach().float() for k in G17_LOSS}
        for i, (kind, t) in enumerate(rec.log):
            if kind != "randn":
                out[f"rng_{kind}"] = t
        loss = 0
        for k in G17_LOSS:
            loss = loss + (T(synth.normal(17, "g17/c/" + k, tuple(res[k].shape))) * res[k]).sum()
        loss.backward()
        out["loss"] = loss.detach()
        for mn, pn in (("nerf_feat", "rgb.0.weight"), ("nerf_feat", "xyz_encoding_1.0.weight"), ("nerf_feat", "beta")):
            gr = dict(models[mn].named_parameters())[pn].grad
            out[f"d_{mn}.{pn}"] = gr if gr is not None else torch.zeros(1)
        save("g17_corr_" + mode, **out)


# --------------------------------------------------------------------------- G18 evaluate_mlp wrapper
G18 = dict(N=7, S=9)


def g18():
    """geom_utils.evaluate_mlp (:19-57) as its callers use it: the coarse net with a materialised per-sample direction
    embedding and a per-ray env code (rendering.py:159-163, chunked), the same with an appearance code (:45-50), sigma_only
    on raw positions (loss_utils.py:122), the skin net with a per-ray (N,1,c) and a shared (1,c) pose code
    (geom_utils.py:228), the feature net on raw positions (rendering.py:176) and the visibility net on an already
    embedded input (rendering.py:378, loss_utils.py:140)."""
    N, S = G18["N"], G18["S"]
    i = {k: T(v) for k, v in synth.evaluate_mlp_inputs(18, G18['N'], G18['S']).items()}
    mp = synth.make_models(18, B=25, with_skin=True, with_feat=True, with_vis=True)
    mp_app = synth.make_models(18, B=0, with_app=True)
    emb, emb_d = nerf.Embedding(3, 10, alpha=10.0), nerf.Embedding(3, 4, alpha=10.0)
    coarse = ref_nerf(mp["coarse"], in_channels_xyz=63, in_channels_dir=27 + 64, init_beta=0.1)
    coarse_app = ref_nerf(mp_app["coarse"], in_channels_xyz=63, in_channels_dir=27 + 64 + 128, init_beta=0.1)
    skin = ref_nerf(mp["nerf_skin"], **NERF_SHAPES["skin"])
    feat = ref_nerf(mp["nerf_feat"], **NERF_SHAPES["feat"])
    vis = ref_nerf(mp["nerf_vis"], **NERF_SHAPES["vis"])
    dir_e = torch.repeat_interleave(emb_d(i["dirs"]), repeats=S, dim=0).view(N, S, -1)      # rendering.py:151, 161
    out = {}
    with torch.no_grad():
        out["coarse"] = geom.evaluate_mlp(coarse, i["xyz"], embed_xyz=emb, dir_embedded=dir_e, code=i["env"], chunk=3)
        out["coarse_app"] = geom.evaluate_mlp(coarse_app, i["xyz"], embed_xyz=emb, dir_embedded=dir_e, code=i["env"][:, None],
                                              appearance_code=i["app"], chunk=4096)
        out["coarse_sigma"] = geom.evaluate_mlp(coarse, i["xyz"], embed_xyz=emb, chunk=N, sigma_only=True)
        out["skin_ray"] = geom.evaluate_mlp(skin, i["xyz"], embed_xyz=emb, code=i["tcode"][:, None], chunk=2)
        out["skin_rest"] = geom.evaluate_mlp(skin, i["xyz"], embed_xyz=emb, code=i["rest"], chunk=8 * 1024)
        out["skin_embedded"] = geom.evaluate_mlp(skin, emb(i["xyz"]), code=i["tcode"], chunk=8 * 1024)    # geom_utils.py:228
        out["feat"] = geom.evaluate_mlp(feat, i["xyz"], embed_xyz=emb, chunk=4096)
        out["vis_embedded"] = geom.evaluate_mlp(vis, emb(i["xyz"]), chunk=5)
    save("g18_evaluate_mlp", **out)


# --------------------------------------------------------------------------- G19 uncertainty head + appearance code
G19_KEYS = ("img_coarse", "sil_coarse", "depth_rnd", "unc_pred", "frame_cyc_dis")
G19_LEAVES = ("rays_o", "rays_d", "bone_rts", "env_code", "appearance_code", "vid_code", "ts", "xysn")
G19_PARAMS = (("nerf_unc", "rgb.0.weight"), ("nerf_unc", "xyz_encoding_1.0.weight"), ("nerf_unc", "dir_encoding.0.weight"),
              ("nerf_unc", "xyz_encoding_8.0.bias"), ("coarse", "dir_encoding.0.weight"), ("coarse", "rgb.0.weight"),
              ("coarse", "sigma.weight"))


def g19():
    """inference_deform with `nerf_unc` (rendering.py:501-516, nerf.py:502-511, moda.py:456-464) and a per-frame
    `appearance_code` feeding the colour branch (rendering.py:369-372, geom_utils.py:45-50, moda.py:263-273):
    outputs in eval mode, outputs + gradients in train mode."""
    N, S, B = 48, 12, 25
    for mode in ("eval", "train"):
        mp = synth.make_models(19, B=B, with_skin=True, perturb_bones=True, with_app=True)
        models, emb = ref_scene(19, B, with_skin=True, perturb_bones=True)
        models["coarse"] = ref_nerf(mp["coarse"], in_channels_xyz=63, in_channels_dir=27 + 64 + 128, init_beta=0.1)
        pu = synth.nerf_params(19, "nerf_unc", D=8, W=256, in_channels_xyz=63, in_channels_dir=32, out_channels=1, init_beta=1.0)
        unc = nerf.NeRFUnc(in_channels_xyz=63, D=8, W=256, out_channels=1, in_channels_dir=32, raw_feat=True, init_beta=1.)
        unc.load_state_dict({k: T(v) for k, v in pu.items()})
        models["nerf_unc"] = unc.eval()
        if mode == "train":
            for m in models.values():
                if isinstance(m, torch.nn.Module):
                    m.train()
        rays = {k: T(v) for k, v in synth.make_rays(19, N, B, rays_per_frame=8, with_app=True).items()}
        rays.update({k: T(v) for k, v in synth.make_unc_rays(19, N, 8).items()})
        if mode == "train":
            for k in G19_LEAVES:
                rays[k].requires_grad_(True)
        with (torch.enable_grad() if mode == "train" else torch.no_grad()):
            res = rendering.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, chunk=1024 * 32, img_size=512,
                                        opts=make_opts())
        out = {k: res[k].detach().float() for k in G19_KEYS}
        if mode == "train":
            loss = 0
            for k in G19_KEYS:
                loss = loss + (T(synth.normal(19, "g19/c/" + k, tuple(res[k].shape))) * res[k]).sum()
            loss.backward()
            out["loss"] = loss.detach()
            for k in G19_LEAVES:
                out["d_" + k] = rays[k].grad
            for mn, pn in G19_PARAMS:
                g = dict(models[mn].named_parameters())[pn].grad
                if g.numel() <= 20000:
                    out[f"d_{mn}.{pn}"] = g
                else:
                    out[f"d_{mn}.{pn}__corner"] = g[:16, :16].clone()
                    out[f"d_{mn}.{pn}__norm"] = g.double().norm()
        save("g19_unc_app_" + mode, **out)


# --------------------------------------------------------------------------- G20 checkpoint layout written by the reference classes
G20 = dict(B=25, offset=[0, 7, 19], t_embed=128, N=32, S=12, frame_ids=[0, 3, 6, 7, 12, 18, 1, 9])


class _RefModel(torch.nn.Module):
    """The parameter-holding attributes of nnutils/moda.py's model, built from the reference's own classes under the
    reference's attribute names (moda.py:186-187 alpha, 230-238 near_far, 253-273 env_code / nerf_coarse, 279-283
    pose_code, 300-302 bones, 315-319 nerf_body_rts, 321-322 skin_aux, 324-332 nerf_skin / rest_pose_code, 344-348
    nerf_vis, 444-449 nerf_feat, 456-464 vid_code / nerf_unc), so that state_dict() has the reference's keys."""

    def __init__(self):
        super().__init__()
        nn = torch.nn
        off = np.asarray(G20["offset"])
        B, td = G20["B"], G20["t_embed"]
        self.alpha = nn.Parameter(torch.Tensor([10.0]))
        self.near_far = nn.Parameter(torch.zeros(int(off[-1]), 2))
        self.env_code = nerf.FrameCode(10, 64, off, scale=1)
        self.nerf_coarse = nerf.NeRF(in_channels_xyz=63, in_channels_dir=27 + 64, init_beta=0.1, enable_semantic=False)
        self.pose_code = nerf.FrameCode(10, td, off)
        self.bones = nn.Parameter(geom.generate_bones(B, B, 0, "cpu"))
        self.nerf_body_rts = nn.Sequential(self.pose_code, nerf.DQ_RTHead(use_quat=True, in_channels_xyz=td, in_channels_dir=0,
                                                                          out_channels=7 * B, raw_feat=True))
        self.skin_aux = nn.Parameter(torch.Tensor([0, 10.0]))
        self.nerf_skin = nerf.NeRF(in_channels_xyz=63 + td, D=5, W=64, in_channels_dir=0, out_channels=B, raw_feat=True,
                                   in_channels_code=td)
        self.rest_pose_code = nn.Embedding(1, td)
        self.nerf_vis = nerf.NeRF(in_channels_xyz=63, D=5, W=64, out_channels=1, in_channels_dir=0, raw_feat=True)
        self.nerf_feat = nerf.NeRF(in_channels_xyz=63, D=5, W=128, out_channels=16, in_channels_dir=0, raw_feat=True,
                                   init_beta=1.)
        self.vid_code = nn.Embedding(len(off) - 1, 32)
        self.nerf_unc = nerf.NeRFUnc(in_channels_xyz=63, D=8, W=256, out_channels=1, in_channels_dir=32, raw_feat=True,
                                     init_beta=1.)


def g20():
    """On-disk formats: the key -> shape map of `model.state_dict()` as train_utils.save_network writes it
    (train_utils.py:292-297), a `vars_<label>.npy` written the way :298-304 does, and what the REFERENCE renders from
    those parameters (per-frame codes from FrameCode / DQ_RTHead, then render_rays) for a handful of frames."""
    m = _RefModel()
    sd = m.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(T(synth.checkpoint_fill(k, tuple(v.shape))))
    m.eval()
    out = {"keys": np.asarray(list(sd.keys())), "shapes": np.asarray([",".join(map(str, v.shape)) for v in sd.values()])}
    B, N, S = G20["B"], G20["N"], G20["S"]
    fid = torch.tensor(G20["frame_ids"])
    rpf = N // len(fid)
    with torch.no_grad():
        bone_rts = m.nerf_body_rts(fid)                                  # (F, 1, B*8)
        tcode = m.pose_code(fid)
        env = m.env_code(fid)
        out["bone_rts"], out["time_embedded"], out["env_code"] = bone_rts.reshape(len(fid), -1), tcode.reshape(len(fid), -1), \
            env.reshape(len(fid), -1)
        rays = {k: T(v) for k, v in synth.make_rays(20, N, 0, rays_per_frame=rpf).items()}
        rep = lambda t: t.reshape(len(fid), -1)[:, None].repeat(1, rpf, 1).reshape(N, -1)     # moda.py:1302-1310
        rays["bone_rts"], rays["time_embedded"], rays["env_code"] = rep(bone_rts), rep(tcode), rep(env)
        rays.update({k: T(v) for k, v in synth.make_unc_rays(20, N, rpf).items()})
        rays["vid_code"] = m.vid_code(torch.tensor([0 if f < G20["offset"][1] else 1 for f in G20["frame_ids"]]))[:, None] \
            .repeat(1, rpf, 1).reshape(N, -1)
        models = {"coarse": m.nerf_coarse, "bones": m.bones, "bones_rst": m.bones.data.clone(), "skin_aux": m.skin_aux,
                  "nerf_skin": m.nerf_skin, "rest_pose_code": m.rest_pose_code, "nerf_vis": m.nerf_vis, "nerf_feat": m.nerf_feat,
                  "nerf_unc": m.nerf_unc}
        emb = {"xyz": nerf.Embedding(3, 10, alpha=10.0), "dir": nerf.Embedding(3, 4, alpha=10.0)}
        res = rendering.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, chunk=1024 * 32, img_size=512,
                                    opts=make_opts(), render_vis=True, obj_bound=np.asarray([0.3, 0.3, 0.3]))
    for k in ("img_coarse", "sil_coarse", "depth_rnd", "xyz_canonical_vis", "frame_cyc_dis", "vis_pred", "unc_pred"):
        out["render_" + k] = res[k]
    out["vid_code"] = rays["vid_code"]
    # vars_<label>.npy exactly as save_network does it: a pickled dict through np.save (bytes kept in the fixture)
    import io
    latest_vars = {"rtk": np.zeros((int(G20["offset"][-1]), 4, 4)), "idk": np.zeros((int(G20["offset"][-1]),)),
                   "obj_bound": np.asarray(0.27), "j2c": np.eye(4)}
    buf = io.BytesIO()
    np.save(buf, latest_vars)
    out["vars_npy_bytes"] = np.frombuffer(buf.getvalue(), np.uint8)
    save("g20_checkpoint", **out)


# --------------------------------------------------------------------------- G21 larger gradient fixture
def g21():
    """The G9 bones+skin gradient check at a size where a single ReLU that switches between two correct fp32 evaluations
    no longer moves a gradient norm visibly: 512 rays x 64 samples (32768 samples; G9 has 576)."""
    N, S, B = 512, 64, 25
    models, emb = ref_scene(21, B, with_skin=True, perturb_bones=True)
    for m in models.values():
        if isinstance(m, torch.nn.Module):
            m.train()
    models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
    models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
    rays = {k: T(v) for k, v in synth.make_rays(21, N, B, rays_per_frame=32).items()}
    for k in GRAD_LEAVES:
        rays[k].requires_grad_(True)
    res = rendering.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, chunk=1024 * 32, img_size=512, opts=make_opts())
    loss = 0
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "frame_cyc_dis"):
        loss = loss + (T(synth.normal(21, "g21/c/" + k, tuple(res[k].shape))) * res[k]).sum()
    loss.backward()
    out = {"loss": loss.detach(), "img_coarse": res["img_coarse"].detach(), "sil_coarse": res["sil_coarse"].detach(),
           "frame_cyc_dis": res["frame_cyc_dis"].detach()}

    def put(key, g):
        if g is None:                           # parameters the path never reads (the sigma head of raw_feat nets)
            return
        if g.numel() <= 4096:
            out[key] = g
        else:                                   # large gradients: a 16x16 corner, 64 strided rows, norm and sum
            g2 = g.reshape(g.shape[0], -1)
            out[key + "__corner"] = g2[:16, :16].clone()
            out[key + "__rows"] = g2[:: max(1, g2.shape[0] // 64)][:64, :64].clone()
            out[key + "__norm"] = g.double().norm()
            out[key + "__sum"] = g.double().sum()

    for k in GRAD_LEAVES:
        put("d_" + k, rays[k].grad)
    for mname in ("coarse", "nerf_skin"):
        for pn, p in models[mname].named_parameters():
            put(f"d_{mname}.{pn}", p.grad)
    put("d_bones_rst", models["bones_rst"].grad)
    put("d_skin_aux", models["skin_aux"].grad)
    put("d_rest_pose_code", models["rest_pose_code"].weight.grad)
    save("g21_grad_large", **out)


# --------------------------------------------------------------------------- G22 per-line pixel files read by the reference
G22 = dict(n_frames=9, img_size=6, W=6, indices=[0, 7, 13, 24, 47, 30])


def g22():
    """Per-line pixel files (preprocess/img2lines.py:33-110) written by moda_amd.pixel_lines.write_pair from synthetic frame
    pairs and READ BY THE REFERENCE'S OWN `utils.io.LineDataset.__getitem__` (utils/io.py:380-454; utils/io.py needs two
    more empty import stubs, `imageio` and `absl`): the items it returns are the fixture moda_amd's LineDataset must
    reproduce from the same files.  (img2lines.py itself pulls in the trainer and cannot be imported; its two helpers
    dict2pix / dict2rtk are three-line dictionary comprehensions restated in pixel_lines.py.)"""
    import importlib
    import tempfile
    sys.modules.setdefault("imageio", types.ModuleType("imageio"))
    if "absl" not in sys.modules:
        absl, fl, app = types.ModuleType("absl"), types.ModuleType("absl.flags"), types.ModuleType("absl.app")
        fl.FLAGS = types.SimpleNamespace()
        for n in ("DEFINE_integer", "DEFINE_string", "DEFINE_bool", "DEFINE_boolean", "DEFINE_float", "DEFINE_enum", "DEFINE_list"):
            setattr(fl, n, lambda *a, **k: None)
        absl.flags, absl.app = fl, app
        sys.modules.update({"absl": absl, "absl.flags": fl, "absl.app": app})
    rio = importlib.import_module("utils.io")
    from moda_amd import pixel_lines as PL
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        seq = os.path.join(tmp, "Pixels", "Full-Resolution", "syn")
        jpg = os.path.join(tmp, "JPEGImages", "Full-Resolution", "syn")
        PL.write_synthetic_sequence(seq, 22, G22["n_frames"], G22["img_size"], G22["W"])
        imglist = [os.path.join(jpg, "%05d.jpg" % i) for i in range(G22["n_frames"])]
        PL.write_synthetic_cameras(os.path.join(tmp, "Cameras", "Full-Resolution", "syn"), 22, G22["n_frames"], skip=(5,))
        ds = rio.LineDataset({"img_size": G22["img_size"]}, imglist=imglist, dataid=3)
        assert len(ds) == (G22["n_frames"] - 1) * G22["img_size"]
        np.random.seed(22)
        for n, idx in enumerate(G22["indices"]):
            elem = ds[idx]
            for k, v in elem.items():
                out[f"item{n}_{k}"] = np.asarray(v)
    save("g22_pixel_lines", **out)


# --------------------------------------------------------------------------- G23 S3IM loss
G23_CASES = {"small": (48, 8), "large": (1100, 4)}       # N < 32*32 (rows repeated) and N > 32*32 (first 1024 rows)


def g23():
    """opts.s3im_loss (off by default, moda.py:170): the stochastic structural-similarity term on the (1,3,32,320) virtual patch
    (rendering.py:528-532, 566-567; loss_utils.py:575-702 S3IM / SSIM / _ssim), with the nine permutations the reference drew
    from torch.randperm recorded.  S3IM.forward masks its arguments IN PLACE (`src_vec *= mask`, `tar_vec *= mask`), so with
    the flag on the reference's result['img_coarse'] and result['img_at_samp'] come back multiplied by sil_at_samp: stored
    as the reference returns them.  Eval outputs, train outputs and gradients."""
    B = 25
    for case, (N, S) in G23_CASES.items():
        for mode in ("eval", "train"):
            models, emb = ref_scene(23, B, with_skin=True, perturb_bones=True)
            if mode == "train":
                models["coarse"].train()
            rays = {k: T(v) for k, v in synth.make_rays(23, N, B, rays_per_frame=4).items()}
            rays.update({k: T(v) for k, v in synth.make_corresp_rays(23, N, B, rays_per_frame=4).items()})
            leaves = ("rays_d", "bone_rts", "time_embedded", "env_code") if case == "small" else ("rays_d",)
            if mode == "train":
                for k in leaves:
                    rays[k].requires_grad_(True)
            perms = []
            orig = torch.randperm

            def randperm(*a, **k):
                t = orig(*a, **k)
                perms.append(t.clone())
                return t
            torch.manual_seed(23)
            torch.randperm = randperm
            try:
                with (torch.enable_grad() if mode == "train" else torch.no_grad()):
                    res = rendering.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, chunk=1024 * 32, img_size=512,
                                                opts=make_opts(dist_corresp=True, use_corresp=True, s3im_loss=True))
            finally:
                torch.randperm = orig
            assert len(perms) == 9 and all(p.shape == (1024,) for p in perms)
            out = {k: res[k].detach().float() for k in ("s3im_loss", "img_coarse", "img_at_samp", "img_loss_samp", "sil_coarse")}
            out["perms"] = torch.stack(perms).to(torch.int32)
            if mode == "train":
                loss = 3.0 * res["s3im_loss"] + (T(synth.normal(23, "g23/c/img", tuple(res["img_loss_samp"].shape)))
                                                 * res["img_loss_samp"]).sum()
                loss.backward()
                out["loss"] = loss.detach()
                for k in leaves:
                    out["d_" + k] = rays[k].grad
                for pn in ("rgb.0.weight", "sigma.weight", "xyz_encoding_1.0.weight", "dir_encoding.0.weight"):
                    out["d_coarse." + pn] = dict(models["coarse"].named_parameters())[pn].grad
                out["d_coarse.beta"] = models["coarse"].beta.grad
            save(f"g23_s3im_{case}_{mode}", **out)


# --------------------------------------------------------------------------- G24 / G25 combined configurations (round 4)
G24 = dict(N=32, S=128, B=25, rays_per_frame=8)
G24_KEYS = G11_KEYS + ("depth_rnd", "xyz_camera_vis", "xyz_canonical_vis")


def g24():
    """BASELINE configs[4] (ama-female) in ONE fixture: hierarchical sampling (N_samples = 128: 64 coarse + 64 importance samples, rendering.py:
    91-114) + the CSE feature head `nerf_feat` (:174-180) + the `feats_at_samp` matching / reprojection / rendered-feature heads
    and the visibility loss (:410-578), i.e. G7 `fine` + G7 `feat` + G11 together.  eval: deterministic depths; train: jittered
    depths and resampling uniforms (perturb = 1), outputs + gradients.  Every random tensor the reference drew is recorded."""
    N, S, B, rpf = G24["N"], G24["S"], G24["B"], G24["rays_per_frame"]
    for mode in ("eval", "train"):
        train = mode == "train"
        models, emb = ref_scene(24, B, with_skin=True, with_feat=True, with_vis=True, perturb_bones=True)
        if train:
            for m in models.values():
                if isinstance(m, torch.nn.Module):
                    m.train()
            models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
            models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
        rays = {k: T(v) for k, v in synth.make_rays(24, N, B, rays_per_frame=rpf).items()}
        rays.update({k: T(v) for k, v in synth.make_corresp_rays(24, N, B, rays_per_frame=rpf).items()})
        rays.update({k: T(v) for k, v in synth.make_feat_rays(24, N, rays_per_frame=rpf).items()})
        leaves = ("rays_o", "rays_d", "bone_rts", "rtk_vec", "time_embedded", "env_code")
        if train:
            for k in leaves:
                rays[k].requires_grad_(True)
        torch.manual_seed(24)
        with RecordRandom() as rec, (torch.enable_grad() if train else torch.no_grad()):
            res = rendering.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, chunk=1024 * 32, img_size=512,
                                        obj_bound=G11_BOUND, use_fine=True, perturb=1.0 if train else 0,
                                        opts=make_opts(dist_corresp=True, use_corresp=True, use_ot=True))
        out = {k: res[k].detach().float() for k in G24_KEYS if k in res}
        for i, (kind, t) in enumerate(rec.log):
            out[f"rng{i}_{kind}"] = t
        if train:
            loss = 0
            for k in G11_LOSS:
                c = T(synth.normal(24, "g24/c/" + k, tuple(res[k].shape) or (1,))).reshape(res[k].shape)
                loss = loss + (c * res[k]).sum()
            loss.backward()
            out["loss"] = loss.detach()
            for k in leaves:
                out["d_" + k] = rays[k].grad
            out["d_bones_rst"] = models["bones_rst"].grad
            for mn, pn in (("nerf_feat", "rgb.0.weight"), ("nerf_feat", "xyz_encoding_1.0.weight"), ("nerf_vis", "rgb.0.weight"),
                           ("coarse", "sigma.weight"), ("coarse", "rgb.0.weight"), ("coarse", "xyz_encoding_5.0.weight"),
                           ("nerf_skin", "rgb.0.weight")):
                out[f"d_{mn}.{pn}"] = dict(models[mn].named_parameters())[pn].grad
            out["d_coarse.beta"] = models["coarse"].beta.grad
        save("g24_cfg5_" + mode, **out)


G25 = dict(N=64, S=32, B=36, rays_per_frame=16)


def g25():
    """BASELINE configs[2] (adult7) in ONE fixture: 36 bones (two 32-bone tiles of the fused warp kernels) + the symmetric-shape
    branch (rendering.py:385-393, the x-flip mask recorded) on 32-sample rays (whole 32-sample groups: the one-kernel skin + warp
    route of the 16-bit modes is taken), i.e. G7 `bones36_skin` + G7 `symm` together; eval, and train outputs + gradients."""
    N, S, B, rpf = G25["N"], G25["S"], G25["B"], G25["rays_per_frame"]
    for mode in ("eval", "train"):
        train = mode == "train"
        models, emb = ref_scene(25, B, with_skin=True, perturb_bones=True)
        if train:
            for m in models.values():
                if isinstance(m, torch.nn.Module):
                    m.train()
            models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
            models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
        rays = {k: T(v) for k, v in synth.make_rays(25, N, B, rays_per_frame=rpf).items()}
        leaves = ("rays_o", "rays_d", "bone_rts", "time_embedded", "env_code")
        if train:
            for k in leaves:
                rays[k].requires_grad_(True)
        torch.manual_seed(25)
        with RecordRandom() as rec, (torch.enable_grad() if train else torch.no_grad()):
            res = rendering.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, chunk=1024 * 32, img_size=512,
                                        opts=make_opts(symm_shape=True))
        out = {k: res[k].detach().float() for k in E2E_KEYS if k in res}
        for i, (kind, t) in enumerate(rec.log):
            out[f"rng{i}_{kind}"] = t
        if train:
            loss = 0
            for k in ("img_coarse", "depth_rnd", "sil_coarse", "frame_cyc_dis"):
                loss = loss + (T(synth.normal(25, "g25/c/" + k, tuple(res[k].shape))) * res[k]).sum()
            loss.backward()
            out["loss"] = loss.detach()
            for k in leaves:
                out["d_" + k] = rays[k].grad
            out["d_bones_rst"] = models["bones_rst"].grad
            out["d_skin_aux"] = models["skin_aux"].grad
            for mn, pn in (("coarse", "sigma.weight"), ("coarse", "xyz_encoding_1.0.weight"), ("nerf_skin", "rgb.0.weight"),
                           ("nerf_skin", "xyz_encoding_1.0.weight")):
                out[f"d_{mn}.{pn}"] = dict(models[mn].named_parameters())[pn].grad
            out["d_coarse.beta"] = models["coarse"].beta.grad
        save("g25_cfg3_" + mode, **out)


# --------------------------------------------------------------------------- G26 feat_match's remaining options (round 4)
G26 = dict(N=12, bound=[0.2, 0.2, 0.2])


def g26():
    """`feat_match(init_pts=..., rt_entropy=True)` (loss_utils.py:297-300, 322-340, 397-402): a lattice of its own around every
    pixel's initial point and the normalised matching entropy.  No caller in the reference passes either (scripts/visualize/
    match.py:101 and loss_utils.py:197 use the defaults); pinned for the drop-in's completeness.  Eval mode (no lattice jitter),
    softmax and Sinkhorn forms."""
    import loss_utils as ref_lu
    N = G26["N"]
    mp = synth.make_models(26, B=25, with_feat=True)
    nerf_feat = ref_nerf(mp["nerf_feat"], **NERF_SHAPES["feat"])
    emb = nerf.Embedding(3, 10, alpha=10.0)
    feats = T(synth.normal(26, "g26/feats", (N, 16)))
    init = T(np.float32(0.05) * synth.normal(26, "g26/init", (N, 3)))
    bound = np.asarray(G26["bound"], np.float32)
    out = {}
    with torch.no_grad():
        for use_ot in (False, True):
            tag = "ot" if use_ot else "softmax"
            p0, u0, _ = ref_lu.feat_match(nerf_feat, emb, feats, bound, use_corr=False, use_ot=use_ot, is_training=False,
                                          rt_entropy=True)
            p1, u1, _ = ref_lu.feat_match(nerf_feat, emb, feats, bound, use_corr=False, use_ot=use_ot, is_training=False,
                                          init_pts=init, rt_entropy=True)
            p2, _ = ref_lu.feat_match(nerf_feat, emb, feats, bound, use_corr=False, use_ot=use_ot, is_training=False, init_pts=init)
            out.update({f"{tag}_pts": p0, f"{tag}_unc": u0, f"{tag}_init_pts": p1, f"{tag}_init_unc": u1, f"{tag}_init_pts_only": p2})
    save("g26_feat_match_options", **out)


# --------------------------------------------------------------------------- float64 truth of the gradient fixtures (round 5)
class _GlobalDraws:
    """The outermost patch of the four torch draw functions for the float64-truth pass.  "record": pass the draw through and
    keep it; "replay": hand back the recorded fp32 draw widened to float64 (shape and kind checked), so that the float64 run
    sees exactly the random numbers of the fp32 run.  The per-case RecordRandom blocks of the generator functions nest inside."""

    def __enter__(self):
        self._orig = (torch.rand, torch.rand_like, torch.randn, torch.randn_like)
        orig = self._orig
        self.pos = 0
        me = self

        def wrap(i, kind):
            def f(*a, **k):
                if F64 == "record":
                    t = orig[i](*a, **k)
                    _DRAWS.append((kind, t.clone()))
                    return t
                assert me.pos < len(_DRAWS), "the float64 run draws more random tensors than the fp32 run"
                kd, t = _DRAWS[me.pos]
                me.pos += 1
                shape = tuple(a[0].shape) if kind.endswith("_like") else tuple(a[0] if isinstance(a[0], (tuple, list, torch.Size)) else a)
                assert kd == kind and tuple(t.shape) == shape, (kd, kind, tuple(t.shape), shape)
                return t.double()
            return f

        torch.rand, torch.rand_like, torch.randn, torch.randn_like = (wrap(0, "rand"), wrap(1, "rand_like"), wrap(2, "randn"),
                                                                       wrap(3, "randn_like"))
        return self

    def __exit__(self, *exc):
        torch.rand, torch.rand_like, torch.randn, torch.randn_like = self._orig


def g64():
    """Float64 TRUTH for the end-to-end gradient fixtures G9, G10, G11, G21, G24, G25.  The fixtures' `d_*`
    arrays are the reference's fp32 autograd output, which is itself only an approximation of the gradient; this pass runs the
    REFERENCE a second time in float64 -- same fp32 input values, same fp32 parameter values, the random draws of the fp32 run
    replayed -- and stores its gradients as <fixture>_f64.npz.  The GPU tests then hold the HIP path to
    rel_l2(hip, f64) <= 2 x rel_l2(reference fp32, f64) + an absolute bar, instead of a loose bar against the fp32 output."""
    global F64
    fns = (g9, g10, g11, g21, g24, g25)
    try:
        for fn in fns:
            del _DRAWS[:]
            F64 = "record"
            with _GlobalDraws():
                fn()
            F64 = "replay"
            torch.set_default_dtype(torch.float64)
            try:
                with _GlobalDraws() as gd:
                    fn()
                assert gd.pos == len(_DRAWS), (fn.__name__, gd.pos, len(_DRAWS))
            finally:
                torch.set_default_dtype(torch.float32)
    finally:
        F64 = None
    print("float64-truth fixtures:", _F64_WRITTEN)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g20",
                                "g21", "g22", "g23", "g24", "g25", "g26"]
    for w in which:
        globals()[w]()
