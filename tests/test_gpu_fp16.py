"""GPU (-m gpu): the fp16 mode (`moda_amd.set_precision("fp16")`, MODA_MLP_F16, round 4) -- fp16 MFMA operands
(v_mfma_f32_32x32x16_f16, the bf16 MFMA's rate, 11 significand bits instead of 8) in `render_rays`' per-sample hot loop: the
8 x 256 colour / density network and the fused skin + warp kernels.  The bar is the north star's own: <= 1e-4 relative against
the REFERENCE's outputs (the fixtures the exact-fp32 mode is held to) and the per-element figure of helpers.elem_err, at the
bf16 mode's speed class.  Nothing saturates silently: the overflow report (`moda_amd.overflow`) is tested here too."""
import numpy as np
import pytest
import torch

import moda_amd
from moda_amd import mlp_pack as mp, overflow, synth
from oracle import moda_oracle as orc
from helpers import E2E_CASES, elem_err, golden, rel_err
from gpu_helpers import T, make_models, make_opts, rays_to_gpu

pytestmark = pytest.mark.gpu


def np_(t):
    return t.detach().cpu().numpy()


@pytest.fixture(autouse=True)
def _no_grad_fp32():
    moda_amd.set_precision("fp32")
    overflow.reset()
    with torch.no_grad():
        yield
    moda_amd.set_precision("fp32")
    overflow.check()                # no test may leave an unreported overflow behind


@pytest.mark.parametrize("name", ["coarse", "skin", "feat", "vis"])
def test_pack_kernel_writes_fp16_stream(name):
    """moda_mlp_pack, mode 3, against the numpy statement of the layout: the bf16 mode's element order, every element the
    round-to-nearest-even fp16 of its weight -- bit for bit (the folded dir layer, an fp32 GEMM on the device and a float64
    product here, to one fp16 ulp)."""
    from test_gpu_parity import _nerf_case
    kw, p, m = _nerf_case(name, seed=13, tag="fused/")
    flags = mp.MLP_F16 | (0 if kw["raw_feat"] else (mp.MLP_SIGMOID | mp.MLP_WITH_SIGMA))
    spec = m._spec(10, flags)
    assert spec.precision == "fp16" and spec.bf16 and not spec.x3
    stream, bias, bd = m._packed(spec, torch.device("cuda:0"))
    idx = mp.stream_index(spec)
    ws_ref, b_ref = idx.pack_numpy(mp.fold_final(p))
    assert stream.dtype == torch.float16 and stream.numel() == ws_ref.shape[0]
    L = moda_amd._lib
    assert stream.numel() * 2 == idx.stream_bytes == L.load().moda_mlp_stream_bytes(
        L._c.byref(L.MlpDesc(W=kw["W"], D=kw["D"], n_out=kw["out_channels"], flags=flags, n_freq=10)))
    # same layout as the bf16 stream of the same network
    idx16 = mp.stream_index(m._spec(10, (flags & ~mp.MLP_F16) | mp.MLP_BF16))
    assert np.array_equal(idx16.widx, idx.widx)
    got, want = np_(stream.float()), orc.f16_round(ws_ref)
    names = mp.weight_names(spec)
    wcode = idx.codes()[0]
    is_dir = (wcode >= 0) & (((wcode >> 24) & 15) == names.index("dir_encoding.0.weight"))
    assert np.array_equal(got[~is_dir], want[~is_dir])
    assert np.abs(got[is_dir] - want[is_dir]).max() <= 2.0 ** -10 * np.abs(want[is_dir]).max()
    assert np.array_equal(np_(bias), b_ref) or np.abs(np_(bias) - b_ref).max() < 1e-6


def test_pack_kernel_writes_split_head_fragments():
    """MODA_MLP_F16_HEADS (the fused fp16 skin + warp kernel): the dir and rgb layers' fragments come as (fp16 rounding, fp16
    residual) pairs, every other layer as in the plain fp16 stream -- against the numpy statement, bit for bit outside the folded
    dir layer (an fp32 product on the device, float64 here: compared as hi + lo sums)."""
    from test_gpu_parity import _nerf_case
    kw, p, m = _nerf_case("skin", seed=13, tag="fused/")
    spec = m._spec(10, mp.MLP_F16 | mp.MLP_F16_HEADS)
    plain = mp.stream_index(m._spec(10, mp.MLP_F16))
    idx = mp.stream_index(spec)
    assert spec.heads_split and idx.part.sum() == 6 and idx.nfrags == plain.nfrags      # (the pairs fit the layers' chunk padding)
    stream, bias, bd = m._packed(spec, torch.device("cuda:0"))
    L = moda_amd._lib
    assert stream.numel() * 2 == idx.stream_bytes == L.load().moda_mlp_stream_bytes(
        L._c.byref(L.MlpDesc(W=kw["W"], D=kw["D"], n_out=kw["out_channels"], flags=mp.MLP_F16 | mp.MLP_F16_HEADS, n_freq=10)))
    ws_ref, _ = idx.pack_numpy(mp.fold_final(p))
    hi = orc.f16_round(ws_ref)
    lo = orc.f16_round((ws_ref - hi).astype(np.float32))
    want = np.where(np.repeat(idx.part.astype(bool), 512), lo, hi)
    got = np_(stream.float())
    names = mp.weight_names(spec)
    wcode = idx.codes()[0]
    is_dir = (wcode >= 0) & (((wcode >> 24) & 15) == names.index("dir_encoding.0.weight"))
    assert np.array_equal(got[~is_dir], want[~is_dir])
    fr = lambda a: a.reshape(-1, 512)
    rows = np.nonzero(idx.part[:-1] == 0)[0]
    rows = rows[idx.part[rows + 1] == 1]                               # fragments followed by their residual fragment
    assert len(rows) == 4 + 2                                          # dir: 1 tile x 2 input tiles x 2 sub-steps; rgb: 1 x 1 x 2
    rec_got, rec_want = fr(got)[rows] + fr(got)[rows + 1], fr(ws_ref)[rows]
    assert np.abs(rec_got - rec_want).max() <= 2.0 ** -20 * np.abs(rec_want).max()      # hi + lo carries 22 bits


def test_pack_kernel_writes_split_rgb_head_of_the_256_wide_network():
    """MODA_MLP_F16_HEADS, W = 256 (`moda_mlp_fwd`): the rgb head's fragments alone come as (rounding, residual) pairs -- 8 pairs
    (1 output tile x 4 input tiles x 2 sub-steps) inside that layer's chunk padding; everything else is the plain fp16 stream."""
    from test_gpu_parity import _nerf_case
    kw, p, m = _nerf_case("coarse", seed=13, tag="fused/")
    flags = mp.MLP_F16 | mp.MLP_F16_HEADS | mp.MLP_SIGMOID | mp.MLP_WITH_SIGMA
    spec = m._spec(10, flags)
    plain = mp.stream_index(m._spec(10, flags & ~mp.MLP_F16_HEADS))
    idx = mp.stream_index(spec)
    assert spec.heads_split and idx.part.sum() == 8 and idx.nfrags == plain.nfrags
    stream, bias, bd = m._packed(spec, torch.device("cuda:0"))
    plain_stream = m._packed(m._spec(10, flags & ~mp.MLP_F16_HEADS), torch.device("cuda:0"))[0]
    L = moda_amd._lib
    assert stream.numel() * 2 == idx.stream_bytes == L.load().moda_mlp_stream_bytes(
        L._c.byref(L.MlpDesc(W=256, D=kw["D"], n_out=3, flags=flags, n_freq=10)))
    ws_ref, _ = idx.pack_numpy(mp.fold_final(p))
    hi = orc.f16_round(ws_ref)
    lo = orc.f16_round((ws_ref - hi).astype(np.float32))
    want = np.where(np.repeat(idx.part.astype(bool), 512), lo, hi)
    got = np_(stream.float())
    names = mp.weight_names(spec)
    wcode = idx.codes()[0]
    is_rgb = (wcode >= 0) & (((wcode >> 24) & 15) == names.index("rgb.0.weight"))
    assert np.array_equal(got[is_rgb], want[is_rgb]) and np.abs(got[is_rgb & np.repeat(idx.part.astype(bool), 512)]).max() > 0
    first = int(np.nonzero(is_rgb)[0][0])                                # the rgb head is the stream's last layer
    assert torch.equal(stream[:first], plain_stream[:first])


def test_split_rgb_head_removes_the_colour_error_of_the_fp16_8x256_network():
    """The 8 x 256 network alone against the fp32 oracle: with single-fp16 heads its colour error is the rgb head's (128 terms,
    nothing behind it but a sigmoid); with the head's weights and activations split (hi + lo, 3 MFMAs per product) the colours
    land several times closer, the density column -- which the head does not touch -- is bit-identical."""
    from moda_amd import nerf
    from test_gpu_parity import _nerf_case
    kw, p, m = _nerf_case("coarse", seed=13, tag="fused/")
    M, n_rows = 256 * 64, 64
    xyz = np.float32(0.35) * synth.normal(13, "hx/xyz", (M, 3))
    dirs = synth.normal(13, "hx/dir", (n_rows, kw["in_channels_dir"]))
    ref = orc.nerf_forward(p, np.concatenate([orc.embedding(xyz, 10, 10.0), np.repeat(dirs, M // n_rows, 0)], -1), D=kw["D"], W=kw["W"],
                           in_channels_xyz=63, in_channels_dir=kw["in_channels_dir"], raw_feat=False)
    out = {}
    for split in (True, False):
        nerf.FP16_SPLIT_HEADS = split
        try:
            out[split] = np_(m.fused(T(xyz), dir_src=T(dirs), precision="fp16"))
        finally:
            nerf.FP16_SPLIT_HEADS = True
    e_on, e_off = rel_err(out[True][:, :3], ref[:, :3]), rel_err(out[False][:, :3], ref[:, :3])
    print(f"fp16 8 x 256 colours vs fp32 oracle: split rgb head {e_on:.2e}, single-fp16 head {e_off:.2e}")
    assert np.array_equal(out[True][:, 3], out[False][:, 3])
    assert e_on < 2e-5 and e_on < e_off / 2.5, (e_on, e_off)
    overflow.check()


def test_one_precision_per_launch():
    L = moda_amd._lib
    for flags in (mp.MLP_F16 | mp.MLP_BF16, mp.MLP_F16 | mp.MLP_BF16X3):
        d = L.MlpDesc(W=256, D=8, n_out=3, flags=flags, n_freq=10)
        assert L.load().moda_mlp_stream_bytes(L._c.byref(d)) == -1
        with pytest.raises(ValueError):
            mp.MlpSpec(W=256, D=8, n_out=3, in_xyz=63, in_dir=91, flags=flags).check()


@pytest.mark.parametrize("name", ["coarse", "skin", "feat", "vis"])
def test_fused_mlp_fp16_matches_fp16_oracle(name):
    """The network alone, fp16 operands forced (`precision="fp16"`): against the oracle that rounds the same operands to fp16 the
    difference is accumulation order, the hardware sine and the folded dir layer (<= 2e-3 of the output's scale); against the
    fp32 oracle it is the fp16 quantisation itself -- 6e-5 for the 8 x 256 network (inside the 1e-4 bar, asserted), 3-6e-4 for
    the raw outputs of the narrow ones (why only `render_rays`' hot loop uses them in fp16: nerf.default_precision)."""
    from test_gpu_parity import _fused_vs_oracle
    e1 = max(_fused_vs_oracle(name, M=64 * 16, n_rows=64, precision="fp16", round_fn=orc.f16_round, tol=2e-3),
             _fused_vs_oracle(name, M=5, n_rows=1, precision="fp16", round_fn=orc.f16_round, tol=2e-3),
             _fused_vs_oracle(name, M=4096 + 3 * 7, n_rows=4096 + 3 * 7, precision="fp16", round_fn=orc.f16_round, tol=2e-3,
                              alpha=6.5, flip=True))
    e2 = _fused_vs_oracle(name, M=64 * 16, n_rows=64, precision="fp16", round_fn=None, tol=1e-4 if name == "coarse" else 2e-3)
    e3 = _fused_vs_oracle(name, M=64 * 16, n_rows=64, precision="bf16", round_fn=None, tol=6e-2)
    if name in ("coarse", "vis"):
        _fused_vs_oracle(name, M=777, n_rows=1, precision="fp16", round_fn=orc.f16_round, tol=2e-3, sigma_only=True)
    print(f"fp16 {name}: vs fp16-rounding oracle {e1:.2e}, vs fp32 oracle {e2:.2e} (bf16 mode: {e3:.2e})")
    assert e2 < e3 / 4          # 3 more significand bits: 8x in expectation


def test_fused_mlp_fp16_many_tiles():
    from test_gpu_parity import _fused_vs_oracle
    _fused_vs_oracle("skin", M=1200 * 256, n_rows=1200, precision="fp16", round_fn=orc.f16_round, tol=2e-3)
    _fused_vs_oracle("coarse", M=300 * 256 + 17, n_rows=1, precision="fp16", round_fn=None, tol=1e-4)


def test_default_precision_of_fp16_mode():
    """Entry points outside render_rays' hot loop return raw network outputs: in fp16 mode they run split-bf16 (G18 at the
    fp32 bars is test_g18_evaluate_mlp_wrapper_matches_reference[fp16])."""
    from moda_amd import nerf
    moda_amd.set_precision("fp16")
    assert nerf.default_precision() == "bf16x3" and nerf.hot_precision() == "fp16"
    with nerf.precision_scope("bf16x3"):
        assert nerf.hot_precision() == "bf16x3"
    assert nerf.get_precision() == "fp16"
    moda_amd.set_precision("bf16")
    assert nerf.default_precision() == nerf.hot_precision() == "bf16"


@pytest.mark.parametrize("name", list(E2E_CASES))
def test_g7_render_rays_fp16_matches_reference_golden(name):
    """All fifteen end-to-end cases against the REFERENCE's outputs in the fp16 mode: <= 1e-4 relative and the per-element bar
    (helpers.elem_err < 1) -- the assertions the exact-fp32 and split-bf16 modes pass."""
    from test_gpu_parity import run_hip_case
    calls = []
    orig = moda_amd.NeRF.fused

    def spy(self, *a, **k):
        calls.append((self.W, k.get("precision")))
        return orig(self, *a, **k)
    moda_amd.NeRF.fused = spy
    try:
        res, g = run_hip_case(name, precision="fp16")
    finally:
        moda_amd.NeRF.fused = orig
    assert (256, "fp16") in calls, "the final pass's 8 x 256 network must have run with fp16 operands"
    worst = (0.0, 0.0, "")
    for k in [k for k in g if not k.startswith("rng")]:
        assert tuple(res[k].shape) == g[k].shape, k
        err, ee = rel_err(np_(res[k]), g[k]), elem_err(np_(res[k]), g[k])
        worst = max(worst, (ee, err, k))
        assert err < 1e-4, (name, k, err)
        assert ee < 1, (name, k, ee)
    print(f"g7 {name} (fp16): worst per-element figure {worst[0]:.3f} (rel {worst[1]:.2e}) on {worst[2]}")
    overflow.check()


def test_g8_cfg1_full_size_fp16():
    """BASELINE config 1 (4096 rays x 64 samples, 25 bones) against the reference's checksum fixture in the fp16 mode; S = 64, so
    both warps run as the fused fp16 skin + warp kernel."""
    g = golden("g8_cfg1")
    models, emb = make_models(0, 25)
    rays = rays_to_gpu(synth.make_rays(0, 4096, 25, rays_per_frame=256))
    warps = []
    orig = moda_amd.NeRF.fused_warp

    def spy(self, *a, **k):
        r = orig(self, *a, **k)
        warps.append(r is not None)
        return r
    moda_amd.NeRF.fused_warp = spy
    moda_amd.set_precision("fp16")
    try:
        res = moda_amd.render_rays(models, emb, rays, N_samples=64, noise_std=0.0, opts=make_opts(), img_size=512)
    finally:
        moda_amd.NeRF.fused_warp = orig
        moda_amd.set_precision("fp32")
    assert warps == [True, True]
    idx = g["ray_index"]
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "xyz_camera_vis", "xyz_canonical_vis", "frame_cyc_dis"):
        a = np_(res[k])
        e, ee = rel_err(a[idx], g[k + "_rays"]), elem_err(a[idx], g[k + "_rays"])
        print(f"g8 {k} (fp16): rel {e:.2e}, per-element figure {ee:.3f}")
        assert e < 1e-4 and ee < 1, (k, e, ee)
        assert abs(a.astype(np.float64).mean() - g[k + "_mean"]) < 1e-4 * max(abs(g[k + "_mean"]), 1e-3), k
    overflow.check()


def test_cfg2_slice_fp16_against_split_bf16():
    """BASELINE config 2's shapes (256 samples per ray, 25 bones; 8192 of the 65 536 rays): the fp16 mode against the split-bf16
    mode (itself <= 1e-6 of exact fp32) -- <= 1e-4 relative on every rendered output (measured: img 1.3e-6 with the split rgb head, depth 9e-6, warped
    points 8e-6) and the per-element figure below 1 on every output (img 0.014, warped points 0.59 with the split heads of the fused
    skin + warp kernel -- 1.18 with single-fp16 heads, `MODA_FP16_HEADS=0`)."""
    N, S = 8192, 256
    models, emb = make_models(0, 25)
    rays = rays_to_gpu(synth.make_rays(1000, N, 25, rays_per_frame=256))
    out = {}
    for prec in ("bf16x3", "fp16"):
        moda_amd.set_precision(prec)
        out[prec] = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "xyz_canonical_vis", "frame_cyc_dis"):
        a, b = np_(out["fp16"][k]), np_(out["bf16x3"][k])
        e, ee = rel_err(a, b), elem_err(a, b)
        print(f"cfg2 slice {k}: fp16 vs bf16x3 rel {e:.2e}, per-element figure {ee:.3f}")
        assert e < 1e-4, (k, e)
        assert ee < 1, (k, ee)
    overflow.check()


def test_fused_fp16_warp_against_two_kernel_split_bf16_route():
    """`moda_mlp_warp_fwd` with fp16 operands against the parity-grade two-kernel route (split-bf16 skin network, exact VALU
    warp), the 8 x 256 network in fp16 on the one side and split-bf16 on the other: warped points within 5e-5 of the scene's
    size, composited cycle distances within the 1e-4 bar; 25 and 36 perturbed bones, both directions."""
    import moda_amd.rendering as R
    for B, seed in ((25, 3), (36, 4)):
        models, emb = make_models(seed, B, perturb_bones=True)
        rays = rays_to_gpu(synth.make_rays(seed, 512, B, rays_per_frame=64))
        res = {}
        for prec, fw in (("fp16", True), ("bf16x3", False)):
            moda_amd.set_precision(prec)
            R.FUSED_WARP = fw
            try:
                res[prec] = moda_amd.render_rays(models, emb, rays, N_samples=64, noise_std=0.0, opts=make_opts(), img_size=512)
            finally:
                R.FUSED_WARP = True
        for k in ("xyz_canonical_vis", "frame_cyc_dis"):
            e = rel_err(np_(res["fp16"][k]), np_(res["bf16x3"][k]))
            print(f"fp16 fused warp, {B} bones, {k}: {e:.2e}")
            assert e < (5e-5 if k == "xyz_canonical_vis" else 1e-4), (B, k, e)


def _skin_case():
    from test_gpu_parity import _nerf_case
    kw, p, m = _nerf_case("skin", seed=13, tag="fused/")
    xyz = T(np.float32(0.35) * synth.normal(13, "ovf/xyz", (4096, 3)))
    code = T(synth.normal(13, "ovf/code", (1, kw["in_channels_xyz"] - 63)))
    return m, xyz, code


def test_overflow_is_reported_not_saturated():
    """An activation beyond fp16's range (a hidden layer scaled by 3e5) and a weight beyond it both raise the flag -- a pinned
    host word the kernels write and the host reads without synchronising -- and the package turns it into Fp16Overflow at the
    next fp16 call / `overflow.check()`; an ordinary launch leaves it alone; the other precisions never touch it."""
    m, xyz, code = _skin_case()
    m.fused(xyz, code=code, precision="fp16")
    torch.cuda.synchronize()
    assert not overflow.tripped()
    m.xyz_encoding_2[0].weight.data.mul_(3e5)
    for prec in ("bf16", "bf16x3", "fp32"):
        m.fused(xyz, code=code, precision=prec)
    torch.cuda.synchronize()
    assert not overflow.tripped()
    out = m.fused(xyz, code=code, precision="fp16")
    torch.cuda.synchronize()
    assert overflow.tripped()
    with pytest.raises(overflow.Fp16Overflow):
        m.fused(xyz, code=code, precision="fp16")         # the next fp16 call reports it before launching anything
    assert not overflow.tripped()                          # ... and clears it
    overflow.reset()
    m.xyz_encoding_2[0].weight.data.mul_(1e3)              # weights ~ 1e7: not representable themselves
    m.fused(xyz, code=code, precision="fp16")
    with pytest.raises(overflow.Fp16Overflow):
        overflow.check()
    # the fused warp kernel reports through the same word
    models, emb = make_models(3, 25)
    rays = rays_to_gpu(synth.make_rays(3, 256, 25, rays_per_frame=64))
    models["nerf_skin"].xyz_encoding_3[0].weight.data.mul_(3e5)
    moda_amd.set_precision("fp16")
    with pytest.raises(overflow.Fp16Overflow):        # raised by a later launch of the same call if the kernel has already run
        moda_amd.render_rays(models, emb, rays, N_samples=64, noise_std=0.0, opts=make_opts(), img_size=512)
        overflow.check()
    torch.cuda.synchronize()
    overflow.reset()


def test_fp16_subnormal_weights_are_not_flushed():
    """A layer whose weights are all fp16 SUBNORMALS (scaled by 1e-4: |w| <= 1.3e-5 < 2^-14) followed by one scaled back: the
    MFMA must take them as they are (error of the fp16 subnormal grid, a few 1e-3) -- flushed to zero the layer would output
    its bias only (error ~1)."""
    m, xyz, code = _skin_case()
    o0 = m.fused(xyz, code=code, precision="fp32")
    m.xyz_encoding_3[0].weight.data.mul_(1e-4)
    m.xyz_encoding_3[0].bias.data.mul_(1e-4)
    m.xyz_encoding_4[0].weight.data.mul_(1e4)
    o1 = m.fused(xyz, code=code, precision="fp32")
    o2 = m.fused(xyz, code=code, precision="fp16")
    assert rel_err(np_(o1), np_(o0)) < 1e-5
    e = rel_err(np_(o2), np_(o1))
    print(f"all-subnormal fp16 layer: {e:.2e} from the fp32 mode")
    assert e < 3e-2
