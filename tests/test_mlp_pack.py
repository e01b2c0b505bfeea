"""CPU: the packed weight stream, replayed through a lane-level MFMA model, reproduces the oracle's NeRF."""
import numpy as np
import pytest

from moda_amd import mlp_pack as mp, synth
from oracle import moda_oracle as orc
from helpers import rel_err
from mfma_model import emulate

CASES = {
    # name: (W, D, n_out, n_code, in_dir, flags)
    "skin_f32": (64, 5, 25, 128, 0, 0),
    "skin_bf16": (64, 5, 25, 128, 0, mp.MLP_BF16),
    "skin36_f32": (64, 5, 36, 128, 0, 0),
    "vis_sigma_only_f32": (64, 5, 1, 0, 0, mp.MLP_SIGMA_ONLY),
    "feat_bf16": (128, 5, 16, 0, 0, mp.MLP_BF16),
    "coarse_f32": (256, 8, 3, 0, 91, mp.MLP_SIGMOID | mp.MLP_WITH_SIGMA),
    "coarse_bf16": (256, 8, 3, 0, 91, mp.MLP_SIGMOID | mp.MLP_WITH_SIGMA | mp.MLP_BF16),
}


@pytest.mark.parametrize("name", list(CASES))
def test_stream_replay_matches_oracle(name):
    W, D, n_out, n_code, in_dir, flags = CASES[name]
    spec = mp.MlpSpec(W=W, D=D, n_out=n_out, in_xyz=63 + n_code, in_dir=in_dir, n_freq=10, flags=flags)
    p = synth.nerf_params(11, name, D=D, W=W, in_channels_xyz=63 + n_code, in_channels_dir=in_dir, out_channels=n_out)
    idx = mp.stream_index(spec)
    assert idx.stream_bytes == idx.nfrags * 1024 and idx.nfrags % spec.chf == 0
    pf = mp.fold_final(p)                                # xyz_encoding_final folded into dir_encoding (no activation between)
    wstream, bias = idx.pack_numpy(pf)
    rnd = orc.bf16_round if spec.bf16 else (lambda a: a)
    wstream = rnd(wstream)
    n = 64
    xyz = np.float32(0.3) * synth.normal(11, name + "/xyz", (n, 3))
    code = synth.normal(11, name + "/code", (n, n_code)) if n_code else np.zeros((n, 0), np.float32)
    dirs = synth.normal(11, name + "/dir", (n, in_dir)) if in_dir else np.zeros((n, 0), np.float32)
    # per-row folded biases, as moda_linear_fwd computes them
    w1, w5, wd = p["xyz_encoding_1.0.weight"], p["xyz_encoding_5.0.weight"], p["dir_encoding.0.weight"]
    rb1 = p["xyz_encoding_1.0.bias"] + code @ w1[:, 63:63 + n_code].T
    rb5 = p["xyz_encoding_5.0.bias"] + code @ w5[:, 63:63 + n_code].T
    rbd = pf["dir_encoding.0.bias"] + dirs @ wd[:, W:W + in_dir].T
    rbd = np.pad(rbd, ((0, 0), (0, spec.NTD * 32 - rbd.shape[1])))
    got = emulate(spec, wstream, bias, xyz, rb1, rb5, rbd, None)
    x = np.concatenate([orc.embedding(xyz, 10), code, dirs], -1)
    ref = orc.nerf_forward(p, x, D=D, W=W, in_channels_xyz=63 + n_code, in_channels_dir=in_dir,
                           raw_feat=not (flags & mp.MLP_SIGMOID), sigma_only=bool(flags & mp.MLP_SIGMA_ONLY),
                           round_fn=None)
    if not (flags & mp.MLP_WITH_SIGMA) and not (flags & mp.MLP_SIGMA_ONLY):
        pass
    elif flags & mp.MLP_SIGMOID:
        pass  # oracle returns [sigmoid(rgb), sigma] already
    tol = 3e-2 if spec.bf16 else 2e-5
    assert got.shape == ref.shape
    assert rel_err(got, ref) < tol, rel_err(got, ref)


def test_pe_slots_cover_every_feature_once():
    seen = {}
    for p in range(32):
        for h in (0, 1):
            f = mp.pe_slot_feature(p, h, 10)
            if f >= 0:
                assert f not in seen
                seen[f] = (p, h)
    assert sorted(seen) == list(range(63))


@pytest.mark.parametrize("bf16", [False, True])
def test_act_features_cover_tile_once(bf16):
    subs, elems = (2, 8) if bf16 else (4, 4)
    seen = set()
    for s in range(subs):
        for j in range(elems):
            for h in (0, 1):
                f = mp.act_feature(bf16, s, j, h)
                assert 0 <= f < 32 and f not in seen
                seen.add(f)
    assert len(seen) == 32


@pytest.mark.parametrize("name", list(CASES))
def test_device_pack_tables_reproduce_the_numpy_pack(name):
    """StreamIndex.codes() -- the (source, offset) tables moda_mlp_pack gathers through on the GPU, with the dir layer read
    from the folded (W/2, W) product instead of dir_encoding's full weight -- yields the stream pack_numpy builds."""
    W, D, n_out, n_code, in_dir, flags = CASES[name]
    spec = mp.MlpSpec(W=W, D=D, n_out=n_out, in_xyz=63 + n_code, in_dir=in_dir, n_freq=10, flags=flags)
    p = synth.nerf_params(11, name, D=D, W=W, in_channels_xyz=63 + n_code, in_channels_dir=in_dir, out_channels=n_out)
    idx = mp.stream_index(spec)
    pf = mp.fold_final(p)
    ws_ref, b_ref = idx.pack_numpy(pf)
    pk = dict(pf)
    pk["dir_encoding.0.weight"] = np.ascontiguousarray(pf["dir_encoding.0.weight"][:, :W])      # the product alone
    ws, b = idx.pack_codes_numpy(pk)
    assert ws.shape == ws_ref.shape and ws.shape[0] % 8 == 0
    assert np.array_equal(ws, ws_ref) and np.array_equal(b, b_ref)
    wcode, bcode = idx.codes()
    assert wcode.dtype == np.int32 and (wcode < 0).sum() == (idx.widx == idx.zero).sum()
