"""GPU (-m gpu): opt-in early ray termination (BASELINE north star: "... exclusive-sum composite and early termination").
The reference composes all S samples of every ray (nnutils/rendering.py:217-221); with opts.early_term_tau = tau > 0 the
HIP path drops the samples whose incoming transmittance is below tau -- at most tau of a ray's weight -- and, with
hierarchical sampling, does not evaluate the 8x256 MLP behind the termination depth found by the coarse pre-pass.
Default (tau = 0) is the reference's arithmetic, bit for bit."""
import numpy as np
import pytest
import torch

from helpers import golden, oracle_scene, rel_err

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import moda_amd
    from moda_amd import synth, rendering as R
    from oracle import moda_oracle as orc
    from gpu_helpers import T, DEV, make_models, make_opts, rays_to_gpu

KEYS = ("img_coarse", "depth_rnd", "sil_coarse", "frame_cyc_dis")


def np_(t):
    return t.detach().cpu().numpy()


@pytest.fixture(autouse=True)
def _no_grad_fp32():
    moda_amd.set_precision("fp32")
    with torch.no_grad():
        yield
    moda_amd.set_precision("fp32")


def test_cfg1_tau_1e4_matches_reference_and_reports_skipped_fraction():
    """BASELINE config 1 (4096 x 64, 25 bones) with tau = 1e-4 against the REFERENCE's outputs (G8): every output within
    1e-4; the fraction of samples the termination drops on this synthetic scene is reported (it is a thin fog: the mean
    transmittance behind the last sample is ~0.04, so nothing terminates)."""
    g = golden("g8_cfg1")
    models, emb = make_models(0, 25)
    rays = rays_to_gpu(synth.make_rays(0, 4096, 25, rays_per_frame=256))
    off = moda_amd.render_rays(models, emb, rays, N_samples=64, noise_std=0.0, opts=make_opts(), img_size=512)
    on = moda_amd.render_rays(models, emb, rays, N_samples=64, noise_std=0.0, opts=make_opts(early_term_tau=1e-4), img_size=512)
    assert "samples_used" not in off and on["samples_used"].dtype == torch.int32
    idx = g["ray_index"]
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "frame_cyc_dis"):
        a = np_(on[k])
        assert rel_err(a[idx], g[k + "_rays"]) < 1e-4, k
        assert abs(a.astype(np.float64).mean() - g[k + "_mean"]) < 1e-4 * max(abs(g[k + "_mean"]), 1e-3), k
    skipped = 1.0 - float(on["samples_used"].float().mean()) / 64
    print(f"cfg1, tau=1e-4: fraction of samples skipped on the synthetic scene = {skipped:.4f}")
    assert 0.0 <= skipped < 0.05


def _dense_scene(seed, B, beta=0.004):
    """The synthetic scene with a 25x sharper SDF->density map (beta 0.1 -> 0.004): rays saturate a few samples after they
    enter the surface, as they do on a trained model."""
    return make_models(seed, B, beta=beta)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_termination_on_an_opaque_scene(precision):
    N, S, B = 2048, 128, 25
    models, emb = _dense_scene(51, B)
    rays_np = synth.make_rays(51, N, B, rays_per_frame=64)
    rays = rays_to_gpu(rays_np)
    moda_amd.set_precision(precision)
    off = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
    on = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, opts=make_opts(early_term_tau=1e-4), img_size=512)
    used = on["samples_used"]
    skipped = 1.0 - float(used.float().mean()) / S
    print(f"opaque scene ({precision}), tau=1e-4: {skipped:.3f} of the samples dropped, {float((used < S).float().mean()):.3f} of the rays terminate")
    assert skipped > 0.2
    for k in KEYS:
        d = float((on[k] - off[k]).abs().max())
        assert d <= 1.5e-4 * max(1.0, float(off[k].abs().max())), (k, d)     # at most tau of each ray's weight is dropped
    if precision == "fp32":
        sc = oracle_scene(51, B)
        sc.coarse = dict(sc.coarse, beta=np.asarray([0.004], np.float32))
        ref = orc.render_rays(sc, {k: v[:256] for k, v in rays_np.items()}, N_samples=S)
        for k in KEYS:
            assert rel_err(np_(on[k])[:256], ref[k]) < 2e-4, (k, rel_err(np_(on[k])[:256], ref[k]))


def test_composite_termination_semantics():
    """moda_composite_fwd with n_live / term_tau: weights are exactly the unterminated ones in front of the cut and exactly
    0 behind it, n_used is the cut, inputs behind n_live are never read (NaN there must not reach any output)."""
    N, S = 37, 200
    rs = torch.rand(N, S, 4, device=DEV)
    rs[..., 3] = (torch.rand(N, S, device=DEV) - 0.45) * 0.4
    z = torch.sort(0.1 + 0.4 * torch.rand(N, S, device=DEV), -1)[0].contiguous()
    rd = torch.randn(N, 3, device=DEV)
    beta = torch.tensor([0.01], device=DEV)
    cyc = torch.rand(N, S, device=DEV)
    full = R.composite(rs, None, z, rd, beta, cyc=cyc)
    tau = 1e-3
    cut = R.composite(rs, None, z, rd, beta, cyc=cyc, term_tau=tau)
    T = full["visibility"]
    want_used = (T >= tau).sum(1).to(torch.int32)
    assert torch.equal(cut["n_used"], want_used)
    keep = torch.arange(S, device=DEV)[None] < want_used[:, None]
    assert torch.equal(cut["weights"], torch.where(keep, full["weights"], torch.zeros_like(full["weights"])))
    assert float((cut["rgb"] - full["rgb"]).abs().max()) <= 1.01 * tau
    # caller-supplied bound + garbage behind it
    n_live = torch.randint(0, S + 1, (N,), device=DEV, dtype=torch.int32)
    n_live[0], n_live[1] = 0, S
    poisoned = rs.clone()
    cyc_p = cyc.clone()
    dead = torch.arange(S, device=DEV)[None] >= n_live[:, None]
    poisoned[dead] = float("nan")
    cyc_p[dead] = float("nan")
    lim = R.composite(poisoned, None, z, rd, beta, cyc=cyc_p, n_live=n_live)
    assert torch.equal(lim["n_used"], n_live)
    for k in ("rgb", "depth", "sil", "weights", "cyc_out"):
        assert torch.isfinite(lim[k]).all(), k
    assert torch.equal(lim["weights"], torch.where(~dead, full["weights"], torch.zeros_like(full["weights"])))
    assert float(lim["rgb"][0].abs().max()) == 0.0 and torch.equal(lim["rgb"][1], full["rgb"][1])


def test_mlp_skips_dead_groups():
    """moda_mlp_live_fwd: 32-sample groups in front of n_live are evaluated exactly as without the bound; the others are
    not written.  Both precisions, all rays dead / all alive / ragged."""
    N, S = 48, 128
    models, emb = make_models(52, 0)
    coarse = models["coarse"]
    xyz = T(np.float32(0.3) * synth.normal(52, "live/xyz", (N, S, 3)))
    dirs = T(synth.normal(52, "live/dir", (N, 91)))
    n_live = torch.randint(0, S + 1, (N,), device=DEV, dtype=torch.int32)
    n_live[:4] = 0
    n_live[4:8] = S
    groups_live = (torch.arange(0, S, 32, device=DEV)[None] < n_live[:, None])           # (N, S/32)
    mask = groups_live.repeat_interleave(32, 1)
    for prec in ("fp32", "bf16"):
        full = coarse.fused(xyz, dir_src=dirs, precision=prec)
        part = coarse.fused(xyz, dir_src=dirs, precision=prec, n_live=n_live)
        assert torch.equal(part[mask], full[mask]), prec
    print(f"mlp skip: {1 - float(groups_live.float().mean()):.2f} of the 32-sample groups not evaluated")


def test_hierarchical_pass_skips_behind_the_termination_depth():
    """use_fine + early_term_tau: the coarse pre-pass gives the termination depth, the final pass's 8x256 MLP skips the
    groups behind it, and the rendered outputs stay within tau of the unterminated render."""
    N, S, B = 1024, 128, 25                       # 64 coarse + 64 importance samples -> 128 merged
    models, emb = _dense_scene(53, B)
    rays = rays_to_gpu(synth.make_rays(53, N, B, rays_per_frame=64))
    seen = []
    orig = moda_amd.NeRF.fused

    def spy(self, *a, **k):
        if k.get("n_live") is not None:
            seen.append(k["n_live"].clone())
        return orig(self, *a, **k)
    moda_amd.NeRF.fused = spy
    try:
        for prec in ("fp32", "bf16"):
            moda_amd.set_precision(prec)
            seen.clear()
            off = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, use_fine=True, opts=make_opts(), img_size=512)
            assert not seen
            on = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, use_fine=True,
                                      opts=make_opts(early_term_tau=1e-4), img_size=512)
            assert len(seen) == 1
            groups = ((seen[0].float() / 32).ceil().clamp(max=S // 32)).mean() / (S // 32)
            print(f"hierarchical ({prec}): the final pass evaluates {float(groups):.3f} of its 32-sample groups; "
                  f"{1 - float(on['samples_used'].float().mean()) / S:.3f} of the samples carry no weight")
            assert float(groups) < 0.8
            for k in KEYS:
                assert torch.isfinite(on[k]).all(), k
                d = float((on[k] - off[k]).abs().max())
                assert d <= 1.5e-4 * max(1.0, float(off[k].abs().max())), (prec, k, d)
    finally:
        moda_amd.NeRF.fused = orig
        moda_amd.set_precision("fp32")


def test_training_route_refuses_early_termination():
    models, emb = make_models(54, 25)
    for m in models.values():
        if isinstance(m, torch.nn.Module):
            m.train()
    rays = rays_to_gpu(synth.make_rays(54, 16, 25, rays_per_frame=8))
    with torch.enable_grad(), pytest.raises(NotImplementedError):
        moda_amd.render_rays(models, emb, rays, N_samples=16, noise_std=0.0, opts=make_opts(early_term_tau=1e-4), img_size=512)
