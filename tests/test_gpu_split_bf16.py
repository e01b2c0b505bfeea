"""GPU (-m gpu): the split-bf16 mode of the fused PE+MLP kernel (`moda_amd.set_precision("bf16x3")`, MODA_MLP_BF16X3) -- every
MFMA operand carried as bf16 hi + lo, three MFMAs per product, exact sincosf encoding: the parity-grade throughput mode.
The bar is the north star's own: <= 1e-4 relative against the REFERENCE's outputs (the fixtures the exact-fp32 mode is held to),
at a third of the bf16 matrix rate instead of a sixteenth."""
import numpy as np
import pytest
import torch

import moda_amd
from moda_amd import mlp_pack as mp, synth
from oracle import moda_oracle as orc
from helpers import E2E_CASES, elem_err, golden, rel_err
from gpu_helpers import T, make_models, make_opts, rays_to_gpu

pytestmark = pytest.mark.gpu


def np_(t):
    return t.detach().cpu().numpy()


@pytest.fixture(autouse=True)
def _no_grad_fp32():
    moda_amd.set_precision("fp32")
    with torch.no_grad():
        yield
    moda_amd.set_precision("fp32")


def x3_round(a):
    """operand as the split mode carries it: bf16(a) + bf16(a - bf16(a))."""
    hi, lo = mp.split_bf16(a, orc.bf16_round)
    return (hi + lo).astype(np.float32)


@pytest.mark.parametrize("name", ["coarse", "skin", "feat", "vis"])
def test_pack_kernel_writes_hi_lo_fragment_pairs(name):
    """moda_mlp_pack in split mode against the numpy statement of the layout: fragment f of the plain stream becomes the
    fragment of the bf16 roundings followed by the fragment of the rounded residuals -- bit for bit."""
    from test_gpu_parity import _nerf_case
    kw, p, m = _nerf_case(name, seed=13, tag="fused/")
    flags = mp.MLP_BF16X3 | (0 if kw["raw_feat"] else (mp.MLP_SIGMOID | mp.MLP_WITH_SIGMA))
    spec = m._spec(10, flags)
    stream, bias, bd = m._packed(spec, torch.device("cuda:0"))
    idx = mp.stream_index(spec)
    pf = mp.fold_final(p)
    ws_ref, b_ref = idx.pack_numpy(pf)
    assert stream.dtype == torch.bfloat16 and stream.numel() == ws_ref.shape[0] and idx.part.sum() > 0
    assert stream.numel() * 2 == idx.stream_bytes == moda_amd._lib.load().moda_mlp_stream_bytes(
        moda_amd._lib._c.byref(moda_amd._lib.MlpDesc(W=kw["W"], D=kw["D"], n_out=kw["out_channels"], flags=flags, n_freq=10)))
    got = np_(stream.float())
    want = mp.stream_x3(idx, ws_ref, orc.bf16_round)
    # the folded dir weights are an fp32 GEMM on the device and a float64 product here (a value on a bf16 rounding boundary may
    # fall either way): every other layer bit for bit, the dir layer as the sums hi + lo its fragment pairs carry
    names = mp.weight_names(spec)
    wcode = idx.codes()[0]
    is_dir = (wcode >= 0) & (((wcode >> 24) & 15) == names.index("dir_encoding.0.weight"))
    assert np.array_equal(got[~is_dir], want[~is_dir])
    fr = lambda a: a.reshape(-1, 512)
    hi_rows = np.nonzero(idx.part[:-1] == 0)[0]
    hi_rows = hi_rows[idx.part[hi_rows + 1] == 1]                    # fragments followed by their residual fragment
    rec_got, rec_want = fr(got)[hi_rows] + fr(got)[hi_rows + 1], fr(want)[hi_rows] + fr(want)[hi_rows + 1]
    assert np.abs(rec_got - rec_want).max() <= 1e-5 * np.abs(rec_want).max()
    assert np.abs(rec_want - fr(ws_ref)[hi_rows]).max() <= 2.0 ** -16 * np.abs(ws_ref).max()      # hi + lo carries 16 bits
    assert np.array_equal(np_(bias), b_ref) or np.abs(np_(bias) - b_ref).max() < 1e-6


@pytest.mark.parametrize("name", ["coarse", "skin", "feat", "vis"])
def test_fused_mlp_split_bf16_matches_fp32_oracle(name):
    """The network alone against the float32 oracle (no rounding hook): <= 2e-5 of the output's scale -- the exact-fp32 mode's
    own bar -- at ragged sizes, per-ray and per-sample code rows, flipped inputs, annealed window, sigma_only."""
    from test_gpu_parity import _fused_vs_oracle
    e = [_fused_vs_oracle(name, M=37 * 16, n_rows=37, precision="bf16x3", round_fn=None, tol=2e-5),
         _fused_vs_oracle(name, M=5, n_rows=1, precision="bf16x3", round_fn=None, tol=2e-5),
         _fused_vs_oracle(name, M=4096 + 3 * 7, n_rows=4096 + 3 * 7, precision="bf16x3", round_fn=None, tol=2e-5, alpha=6.5,
                          flip=True)]
    if name in ("coarse", "vis"):
        e.append(_fused_vs_oracle(name, M=777, n_rows=1, precision="bf16x3", round_fn=None, tol=2e-5, sigma_only=True))
    e16 = _fused_vs_oracle(name, M=37 * 16, n_rows=37, precision="bf16", round_fn=None, tol=6e-2)
    print(f"split-bf16 {name}: worst {max(e):.2e} vs the fp32 oracle (plain bf16 mode: {e16:.2e})")


def test_fused_mlp_split_bf16_many_tiles():
    from test_gpu_parity import _fused_vs_oracle
    _fused_vs_oracle("skin", M=1200 * 256, n_rows=1200, precision="bf16x3", round_fn=None, tol=2e-5)
    _fused_vs_oracle("coarse", M=300 * 256 + 17, n_rows=1, precision="bf16x3", round_fn=None, tol=2e-5)


@pytest.mark.parametrize("name", list(E2E_CASES))
def test_g7_render_rays_split_bf16_matches_reference_golden(name):
    """All fifteen end-to-end cases against the REFERENCE's outputs in the split-bf16 mode: <= 1e-4 relative and the per-element
    bar (helpers.elem_err < 1), the same assertions the exact-fp32 mode passes."""
    from test_gpu_parity import run_hip_case
    res, g = run_hip_case(name, precision="bf16x3")
    worst = (0.0, 0.0, "")
    for k in [k for k in g if not k.startswith("rng")]:
        assert tuple(res[k].shape) == g[k].shape, k
        err, ee = rel_err(np_(res[k]), g[k]), elem_err(np_(res[k]), g[k])
        worst = max(worst, (ee, err, k))
        assert err < 1e-4, (name, k, err)
        assert ee < 1, (name, k, ee)
    print(f"g7 {name} (bf16x3): worst per-element figure {worst[0]:.3f} (rel {worst[1]:.2e}) on {worst[2]}")


def test_g8_cfg1_full_size_split_bf16():
    """BASELINE config 1 (4096 rays x 64 samples, 25 bones) against the reference's checksum fixture in the split-bf16 mode."""
    g = golden("g8_cfg1")
    models, emb = make_models(0, 25)
    rays = rays_to_gpu(synth.make_rays(0, 4096, 25, rays_per_frame=256))
    moda_amd.set_precision("bf16x3")
    try:
        with torch.no_grad():
            res = moda_amd.render_rays(models, emb, rays, N_samples=64, noise_std=0.0, opts=make_opts(), img_size=512)
    finally:
        moda_amd.set_precision("fp32")
    idx = g["ray_index"]
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "xyz_camera_vis", "xyz_canonical_vis", "frame_cyc_dis"):
        a = np_(res[k])
        e, ee = rel_err(a[idx], g[k + "_rays"]), elem_err(a[idx], g[k + "_rays"])
        print(f"g8 {k} (bf16x3): rel {e:.2e}, per-element figure {ee:.3f}")
        assert e < 1e-4 and ee < 1, (k, e, ee)
        assert abs(a.astype(np.float64).mean() - g[k + "_mean"]) < 1e-4 * max(abs(g[k + "_mean"]), 1e-3), k
