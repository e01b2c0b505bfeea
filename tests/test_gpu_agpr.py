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


def _run(m, xyz, dirs, agpr, **kw):
    old = os.environ.get("MODA_MLP_AGPR")
    os.environ["MODA_MLP_AGPR"] = "1" if agpr else "0"
    try:
        with torch.no_grad():
            return m.fused(xyz, dir_src=dirs, precision="bf16", **kw)
    finally:
        if old is None:
            os.environ.pop("MODA_MLP_AGPR", None)
        else:
            os.environ["MODA_MLP_AGPR"] = old


@pytest.mark.parametrize("N,S", [(1, 32), (7, 32), (513, 64), (4096, 96), (1000, 256), (8192, 256)])
def test_agpr_form_is_bit_identical_to_the_eight_wave_form(N, S):
    m = _net()
    xyz = T(np.float32(0.3) * synth.normal(61, "ag/xyz", (N, S, 3)))
    dirs = T(synth.normal(61, "ag/dir", (N, 91)))
    a, b = _run(m, xyz, dirs, False), _run(m, xyz, dirs, True)
    assert torch.isfinite(b).all() and torch.equal(a, b), float((a - b).abs().max())
    # the sigma-only pre-pass form (hierarchical sampling) takes the same kernel
    a, b = _run(m, xyz, dirs, False, sigma_only=True), _run(m, xyz, dirs, True, sigma_only=True)
    assert torch.equal(a, b)


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


def test_render_rays_is_bit_identical_under_both_forms():
    models, emb = make_models(0, 25)
    rays = rays_to_gpu(synth.make_rays(1000, 8192, 25, rays_per_frame=256))
    moda_amd.set_precision("bf16")
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
