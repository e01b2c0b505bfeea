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


def _truth_views(name, a, g):
    """The stored views of gradient `name` in fixture g evaluated on the full tensor a: [(view, got, stored)].  Small tensors
    are stored whole; large ones as a 16 x 16 corner (+ 64 strided rows in G21) + norm + sum (tests/golden/gen_golden.py `put`)."""
    if name in g:
        return [("all", a, g[name])]
    if name + "__norm" not in g:
        return []
    a2 = a.reshape(a.shape[0], -1)
    views = [("corner", a2[:16, :16], g[name + "__corner"])]
    if name + "__rows" in g:
        views.append(("rows", a2[:: max(1, a2.shape[0] // 64)][:64, :64], g[name + "__rows"]))
    views.append(("norm", np.linalg.norm(a.astype(np.float64)), g[name + "__norm"]))
    views.append(("sum", (a.astype(np.float64).sum(), a.size), g[name + "__sum"]))
    return views


def f64_truth_table(fixture, got):
    """Per gradient tensor and stored view: (name, view, e_got, e_ref) with e_x = distance of x from the float64 TRUTH --
    the reference itself run in float64 on the same inputs and random draws (tests/golden/<fixture>_f64.npz, gen_golden.py g64) --
    for x = the gradients under test (`got`: name -> array) and x = the reference's own fp32 autograd output (<fixture>.npz).
    Distance: relative L2 for tensors / corners / rows, |dx| / x for the norm, and |d sum| / (sqrt(n) norm) for the sum of the n
    entries (a sum of signed entries cancels, so relative to itself its error is unbounded; by Cauchy-Schwarz this figure is a
    lower bound of the tensor's relative L2 error, so the same bars apply)."""
    g32, g64 = golden(fixture), golden(fixture + "_f64")
    rows = []
    for name, a in got.items():
        a = np.asarray(a, np.float64)
        if name not in g64 and name + "__norm" not in g64:
            continue
        stored64 = _truth_views(name, a, g64)
        for view, x, t in stored64:
            r = g32[name if view == "all" else f"{name}__{view}"].astype(np.float64)
            if view == "sum":
                x, n = x
                scale = float(g64[name + "__norm"]) * np.sqrt(n)
                e_got, e_ref = abs(float(x) - float(t)) / scale, abs(float(r) - float(t)) / scale
            elif view == "norm":
                e_got, e_ref = abs(float(x) - float(t)) / float(t), abs(float(r) - float(t)) / float(t)
            else:
                e_got, e_ref = rel_l2(x, t), rel_l2(r, t)
            rows.append((name, view, e_got, e_ref))
    return rows


TRUTH_FACTOR = 2.0      # (i)  no further from the float64 truth than twice the reference's own fp32 autograd is ...
TRUTH_FLOOR = 2e-5      #      ... or than this floor where the reference sits at round-off level
TRUTH_BAR = 1e-3        # (ii) absolute bar on every gradient tensor ...
TRUTH_BAR_ILL = 3e-3    #      ... except where the reference's OWN fp32 output misses half of it: max(this, factor * e_ref), printed
AMB_TOL = 2e-6          # gen_golden.py AMB_TOL: relative margin below which a ReLU's state is undetermined at fp32 accuracy
FLIP_GAIN = 16.0        # one sample whose ReLU flips moves a parameter gradient by <= FLIP_GAIN / (samples of the fixture) of its norm
PER_RAY = ("d_rays_o", "d_rays_d", "d_bone_rts", "d_bone_rts_target", "d_time_embedded", "d_env_code", "d_rtk_vec", "d_rtk_vec_target")


def certified_rays(fixture, n_rays, tol=AMB_TOL):
    """The conditioning certificate of a *_f64 fixture (gen_golden.py _watch_relus: computed by the reference itself in float64):
    rays holding a sample with a ReLU pre-activation closer to zero than tol (relative to its layer's largest) -- there two
    correct fp32 evaluations may take different sides, and the gradient of that ray is not determined at fp32 accuracy -- and
    the number of such rows in calls that are not ray-major (the 20^3 matching lattice: it feeds every ray).
    -> (sorted ray list, other rows, samples per fixture)."""
    g = golden(fixture + "_f64")
    rows, mg = g["amb_rows"], g["amb_margins"]
    rays, other, m_max = set(), 0, 0
    for (n, r), m in zip(rows.tolist(), mg.tolist()):
        ray_major = n % n_rays == 0 and n != 8000
        if ray_major:
            m_max = max(m_max, n)
        if m >= tol:
            continue
        if ray_major:
            rays.add(r * n_rays // n)
        else:
            other += 1
    return sorted(rays), other, m_max


def assert_gradients_within_f64_truth(fixture, got, label, n_rays, n_samples, factor=TRUTH_FACTOR, floor=TRUTH_FLOOR,
                                      bar=TRUTH_BAR, bar_ill=TRUTH_BAR_ILL):
    """gradients are held against the float64 TRUTH (the reference run in float64, <fixture>_f64.npz), not
    against the reference's fp32 output at a loose bar.  e_x below = distance from the truth (f64_truth_table).

    A. per-ray gradients (rays_o, rays_d, bone_rts, time_embedded, env_code, rtk_vec ...) on the rays the fixture's certificate
       leaves (no ReLU within AMB_TOL of zero in float64): e_got <= max(factor * e_ref, floor) AND e_got <= bar -- immune to ReLU
       coin flips, so this is the tight statement about the whole backward chain.
    B. every tensor, all rays: e_got <= bar (max(bar_ill, factor * e_ref) where the reference's own fp32 output is further than
       bar / 2 from the truth; printed) and e_got <= max(factor * e_ref, FLIP_GAIN / samples): no further from the truth than the
       reference's own fp32 autograd, up to what ONE flipped sample is worth.
    C. a violation of B is accepted only when it is a certified coin flip: the deviation of the per-ray tensors sits on
       certified rays F (removing them restores B's bound), and then parameter gradients may be FLIP_GAIN * |F| / samples off;
       printed.  Anything else fails.
    The table of every view lands in gpurun_out/grad_truth/<fixture>__<label>.json."""
    import json
    rows = f64_truth_table(fixture, got)
    assert rows, fixture
    cert, other, _ = certified_rays(fixture, n_rays)
    keep = np.setdiff1d(np.arange(n_rays), np.asarray(cert, np.int64))
    g32, g64 = golden(fixture), golden(fixture + "_f64")
    flipf = FLIP_GAIN / float(n_rays * n_samples)

    def per_ray_err(name, rays_):
        a = np.asarray(got[name], np.float64)[rays_]
        return rel_l2(a, g64[name][rays_]), rel_l2(g32[name].astype(np.float64)[rays_], g64[name][rays_])

    out_dir = os.path.join(os.path.dirname(GOLDEN.rstrip("/")), "..", "gpurun_out", "grad_truth")
    try:
        os.makedirs(out_dir, exist_ok=True)
        json.dump({"views": [dict(name=n, view=v, e_got=eg, e_ref=er) for n, v, eg, er in rows], "certified_rays": cert,
                   "certified_rows_not_ray_major": other, "rays": n_rays, "samples_per_ray": n_samples},
                  open(os.path.join(out_dir, f"{fixture}__{label}.json"), "w"), indent=0)
    except OSError:
        pass
    # ---- A: per-ray tensors on the uncertified rays
    tight = []
    for name in got:
        if name in PER_RAY and name in g64 and g64[name].shape[0] == n_rays and len(keep):
            eg, er = per_ray_err(name, keep)
            tight.append((name, eg, er))
            assert eg <= max(factor * er, floor) and eg <= bar, (fixture, label, "uncertified rays", name, eg, er)
    # ---- B / C
    worst = max(rows, key=lambda r: r[2])
    ratios = [r[2] / max(r[3], floor / factor) for r in rows]
    print(f"{fixture} [{label}] vs float64 truth: {len(rows)} views, worst e_got {worst[2]:.2e} ({worst[0]}:{worst[1]}, reference fp32 "
          f"{worst[3]:.2e}); median e_got {np.median([r[2] for r in rows]):.2e} / reference {np.median([r[3] for r in rows]):.2e}, "
          f"median ratio {np.median(ratios):.2f}; {len(keep)} of {n_rays} rays uncertified"
          + (f", there worst per-ray tensor {max(tight, key=lambda t: t[1])[1]:.2e} (reference {max(tight, key=lambda t: t[1])[2]:.2e})" if tight else ""))
    bad = []
    for n, v, eg, er in rows:
        ill = er > bar / 2
        if ill:
            print(f"   ill-conditioned entry {n}:{v}: the reference's own fp32 autograd is {er:.2e} from the truth; here {eg:.2e}")
        if eg > max(factor * er, flipf, floor) or eg > (max(bar_ill, factor * er) if ill else bar):
            bad.append((n, v, eg, er))
    if not bad:
        assert np.median(ratios) <= factor, (fixture, label, "median ratio", float(np.median(ratios)))
        return rows
    # C: certified coin flips.  F = certified rays on which a per-ray tensor deviates by more than the bar allows
    F = set()
    for name in got:
        if name in PER_RAY and name in g64 and g64[name].shape[0] == n_rays:
            a, t = np.asarray(got[name], np.float64).reshape(n_rays, -1), g64[name].reshape(n_rays, -1)
            dev = np.linalg.norm(a - t, axis=1) / max(np.linalg.norm(t), 1e-300)
            F.update(int(r) for r in cert if dev[r] > max(floor, bar / np.sqrt(n_rays)))
    assert F, (fixture, label, "violations without a certified coin flip", bad)
    rest = np.setdiff1d(np.arange(n_rays), np.asarray(sorted(F), np.int64))
    print(f"   CERTIFIED COIN FLIP on ray(s) {sorted(F)} (float64 margin of their closest ReLU < {AMB_TOL:g}): {len(bad)} view(s) beyond the bar, "
          f"parameter gradients allowed {FLIP_GAIN * len(F) / float(n_rays * n_samples):.1e}")
    for n, v, eg, er in bad:
        if n in PER_RAY and v == "all":
            eg2, er2 = per_ray_err(n, rest)
            assert eg2 <= max(factor * er2, flipf, floor) and eg2 <= bar, (fixture, label, "beyond the flipped rays", n, eg2, er2)
        else:
            assert eg <= max(FLIP_GAIN * len(F) / float(n_rays * n_samples), factor * er), (fixture, label, "parameter gradient", n, v, eg, er)
    return rows


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
