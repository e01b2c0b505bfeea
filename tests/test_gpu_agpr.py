"""GPU (-m gpu): the four-wave, two-column-block form of the 8 x 256 bf16 kernel with the activations in asm-owned AGPRs (PrecBF16A,
the default since round 5) against the eight-wave one-block form of rounds 2-4 (MODA_MLP_AGPR=0): the same arithmetic per sample, so
BIT-identical outputs -- at config 2's size, at ragged and short sizes, through the whole render_rays, and launch after launch (the
three asm-MFMA hazards met while building it all showed as outputs that changed from launch to launch)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import moda_amd
    from moda_amd import synth
    from gpu_helpers import T, make_models, make_opts, nerf_from_params, rays_to_gpu

KW = dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False)


def _net():
    return nerf_from_params(synth.nerf_params(5, "mb/coarse", D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3), **KW)


def _run(m, xyz, dirs, agpr, precision="bf16", **kw):
    old = os.environ.get("MODA_MLP_AGPR")
    os.environ["MODA_MLP_AGPR"] = "1" if agpr else "0"
    try:
        with torch.no_grad():
            return m.fused(xyz, dir_src=dirs, precision=precision, **kw)
    finally:
        if old is None:
            os.environ.pop("MODA_MLP_AGPR", None)
        else:
            os.environ["MODA_MLP_AGPR"] = old


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("N,S", [(1, 32), (7, 32), (513, 64), (4096, 96), (1000, 256), (8192, 256)])
def test_agpr_form_is_bit_identical_to_the_eight_wave_form(N, S, precision):
    """bf16: PrecBF16A; fp16: PrecF16A -- the parity-grade mode's kernel, whose rgb head runs split (weights and activations as
    fp16 rounding + residual, three MFMAs; the residual tiles live in AGPRs too) and whose overflow probe rides in the asm."""
    m = _net()
    xyz = T(np.float32(0.3) * synth.normal(61, "ag/xyz", (N, S, 3)))
    dirs = T(synth.normal(61, "ag/dir", (N, 91)))
    a, b = _run(m, xyz, dirs, False, precision), _run(m, xyz, dirs, True, precision)
    assert torch.isfinite(b).all() and torch.equal(a, b), float((a - b).abs().max())
    if precision == "fp16":
        moda_amd.overflow.check()
    # the sigma-only pre-pass form (hierarchical sampling) takes the same kernel
    a, b = _run(m, xyz, dirs, False, precision, sigma_only=True), _run(m, xyz, dirs, True, precision, sigma_only=True)
    assert torch.equal(a, b)


def test_fp16_agpr_form_reports_an_overflow_like_the_eight_wave_form():
    """An activation beyond fp16's range must raise the overflow word in the AGPR form too (the probe is an asm statement there)."""
    m = _net()
    with torch.no_grad():
        m.xyz_encoding_3[0].weight.mul_(1.0e6)
    xyz = T(np.float32(0.3) * synth.normal(63, "ag/xyz", (64, 64, 3)))
    dirs = T(synth.normal(63, "ag/dir", (64, 91)))
    for ag in (False, True):
        moda_amd.overflow.reset() if hasattr(moda_amd.overflow, "reset") else None
        _run(m, xyz, dirs, ag, "fp16")
        with pytest.raises(moda_amd.overflow.Fp16Overflow):
            moda_amd.overflow.check()


def test_agpr_form_repeats_bit_for_bit_at_full_size():
    """65 536 x 256 (config 2), twelve launches of the AGPR form: every one equal to the first and to the eight-wave form."""
    N, S = 65536, 256
    m = _net()
    xyz = T(np.float32(0.3) * synth.normal(62, "ag/xyz", (4096 * 16, 3))).repeat(N * S // (4096 * 16), 1).view(N, S, 3).contiguous()
    dirs = T(synth.normal(62, "ag/dir", (N, 91)))
    ref = _run(m, xyz, dirs, False)
    for i in range(12):
        b = _run(m, xyz, dirs, True)
        assert torch.equal(ref, b), (i, float((ref - b).abs().max()))


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_render_rays_is_bit_identical_under_both_forms(precision):
    models, emb = make_models(0, 25)
    rays = rays_to_gpu(synth.make_rays(1000, 8192, 25, rays_per_frame=256))
    moda_amd.set_precision(precision)
    try:
        out = []
        for ag in ("0", "1"):
            os.environ["MODA_MLP_AGPR"] = ag
            with torch.no_grad():
                out.append(moda_amd.render_rays(models, emb, rays, N_samples=256, perturb=0, noise_std=0.0, opts=make_opts(), img_size=512))
    finally:
        os.environ.pop("MODA_MLP_AGPR", None)
        moda_amd.set_precision("fp32")
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "xyz_canonical_vis", "frame_cyc_dis"):
        assert torch.equal(out[0][k], out[1][k]), k
