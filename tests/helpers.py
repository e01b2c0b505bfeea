"""Shared test helpers: golden loading and oracle scene construction from synth parameters."""
import os

import numpy as np

from moda_amd import synth
from oracle import moda_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def cast(d, dt):
    if isinstance(d, dict):
        return {k: cast(v, dt) for k, v in d.items()}
    if isinstance(d, np.ndarray) and d.dtype.kind == "f":
        return d.astype(dt)
    return d


def oracle_scene(seed, B, with_skin=True, with_feat=False, with_vis=False, alpha=10.0, perturb_bones=False,
                 dtype=np.float32, with_dis=False):
    mp = cast(synth.make_models(seed, B=B, with_skin=with_skin, with_feat=with_feat, with_vis=with_vis,
                                perturb_bones=perturb_bones, with_dis=with_dis), dtype)
    return orc.Scene(mp["coarse"], bones_rst=mp.get("bones_rst"), skin_aux=mp.get("skin_aux"),
                     nerf_skin=mp.get("nerf_skin"), rest_pose_code=mp.get("rest_pose_code"),
                     nerf_feat=mp.get("nerf_feat"), nerf_vis=mp.get("nerf_vis"), alpha_xyz=alpha, alpha_dir=alpha,
                     nerf_dis=mp.get("nerf_dis"))


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the 'rel' of the 1e-4 rel fp32 bar (scale = the tensor's own magnitude)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def elem_err(a, b, rtol=1e-4, afrac=1e-5):
    """Per-element figure beside rel_err: max over elements of |a-b| / (rtol |b| + afrac max|b|); < 1 means every
    element is within rtol of its own magnitude, up to an absolute floor of afrac of the tensor's largest magnitude
    (fp32 round-off of sums whose terms are of the tensor's scale cannot be relative to an element that is ~0)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if b.size == 0:
        return 0.0
    return float((np.abs(a - b) / (rtol * np.abs(b) + afrac * max(np.abs(b).max(), 1e-30))).max())


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def unc_scene_params(seed):
    """Parameters of the G19 scene: coarse net with the 128-wide appearance code, skin net, uncertainty head."""
    mp = synth.make_models(seed, B=25, with_skin=True, perturb_bones=True, with_app=True)
    mp["nerf_unc"] = synth.nerf_params(seed, "nerf_unc", D=8, W=256, in_channels_xyz=63, in_channels_dir=32, out_channels=1,
                                       init_beta=1.0)
    return mp


def checkpoint_states(g):
    """The reference-keyed state dict of the G20 fixture (key -> shape map written by the reference's classes; values
    from synth.checkpoint_fill, the same the generator loaded into the reference model), as numpy arrays."""
    out = {}
    for k, sh in zip(g["keys"].tolist(), g["shapes"].tolist()):
        shape = tuple(int(v) for v in sh.split(",")) if sh else ()
        out[k] = synth.checkpoint_fill(k, shape)
    return out


# end-to-end cases shared by the oracle-vs-golden and HIP-vs-oracle tests: name -> kwargs
E2E_CASES = {
    "nobones": dict(B=0),
    "bones_noskin": dict(B=25, with_skin=False),
    "bones_skin": dict(B=25),
    "bones36_skin": dict(B=36, perturb_bones=True),
    "alpha65": dict(B=25, alpha=6.5),
    "perturb": dict(B=25, perturb=1.0),
    "symm": dict(B=25, symm=True),
    "fine": dict(B=25, use_fine=True, S=32),
    "fine_perturb_symm": dict(B=25, use_fine=True, S=32, perturb=1.0, symm=True),
    "feat": dict(B=25, with_feat=True),
    "render_vis": dict(B=25, with_vis=True, render_vis=True, obj_bound=[0.15, 0.15, 0.15]),
    "disp": dict(B=25, use_disp=True),
    "rgb_filter": dict(B=25, rgb_filter=True),
    "dis": dict(B=25, with_dis=True),
    "dis_fine": dict(B=25, with_dis=True, use_fine=True, S=32),
}


def e2e_random_inputs(g, case):
    """Map the RNG log stored in a g7 fixture onto the oracle's injected-randomness arguments.

    Draw order in the reference (SURVEY.md 8a note 9): rand(N,S) if perturb; [pre-pass: rand_like symm, randn];
    rand(N,S/2) pdf if perturb; rand_like symm; randn.  randn tensors are scaled by noise_std by the caller.
    """
    log = sorted((k for k in g if k.startswith("rng")), key=lambda k: int(k[3:].split("_")[0]))
    seq = [(k.split("_", 1)[1], g[k]) for k in log]
    kw = {}
    i = 0
    perturb = case.get("perturb", 0) > 0
    if perturb:
        kw["perturb_rand"] = seq[i][1]; i += 1
    if case.get("use_fine"):
        if case.get("symm"):
            kw["symm_mask_pre"] = seq[i][1] < 0.5; i += 1
        kw["noise_pre_raw"] = seq[i][1]; i += 1
        if perturb:
            kw["pdf_u"] = seq[i][1]; i += 1
    if case.get("symm"):
        kw["symm_mask"] = seq[i][1] < 0.5; i += 1
    kw["noise_raw"] = seq[i][1]; i += 1
    assert i == len(seq)
    return kw
