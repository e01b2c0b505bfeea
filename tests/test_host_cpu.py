"""CPU (no GPU needed): the C-ABI library loads and exports every symbol include/moda_hip.h declares, the host
packer agrees with the library's own stream-size arithmetic, and the product path refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import moda_amd
from moda_amd import _lib, build, mlp_pack as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.load()


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "moda_hip.h")).read()
    declared = set(re.findall(r"\b(moda_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in moda_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.moda_abi_version() == _lib.ABI_VERSION == 9


SPECS = [
    dict(W=256, D=8, n_out=3, in_xyz=63, in_dir=91, flags=mp.MLP_SIGMOID | mp.MLP_WITH_SIGMA),
    dict(W=256, D=8, n_out=3, in_xyz=63, in_dir=91, flags=mp.MLP_SIGMOID | mp.MLP_WITH_SIGMA | mp.MLP_BF16),
    dict(W=256, D=8, n_out=3, in_xyz=63, in_dir=91, flags=mp.MLP_SIGMA_ONLY),
    dict(W=64, D=5, n_out=25, in_xyz=191, in_dir=0, flags=0),
    dict(W=64, D=5, n_out=36, in_xyz=191, in_dir=0, flags=mp.MLP_BF16),
    dict(W=128, D=5, n_out=16, in_xyz=63, in_dir=0, flags=mp.MLP_BF16),
    dict(W=64, D=5, n_out=1, in_xyz=63, in_dir=0, flags=mp.MLP_SIGMOID),
    # split-bf16: (hi, lo) fragment pairs, padded per layer after pairing
    dict(W=256, D=8, n_out=3, in_xyz=63, in_dir=91, flags=mp.MLP_SIGMOID | mp.MLP_WITH_SIGMA | mp.MLP_BF16X3),
    dict(W=256, D=8, n_out=3, in_xyz=63, in_dir=91, flags=mp.MLP_SIGMA_ONLY | mp.MLP_BF16X3),
    dict(W=64, D=5, n_out=36, in_xyz=191, in_dir=0, flags=mp.MLP_BF16X3),
    dict(W=128, D=5, n_out=16, in_xyz=63, in_dir=0, flags=mp.MLP_BF16X3),
    dict(W=64, D=5, n_out=1, in_xyz=63, in_dir=0, flags=mp.MLP_BF16X3),
    dict(W=256, D=8, n_out=3, in_xyz=63, in_dir=91, flags=mp.MLP_F16 | mp.MLP_SIGMOID | mp.MLP_WITH_SIGMA),
    dict(W=128, D=5, n_out=16, in_xyz=63, in_dir=0, flags=mp.MLP_F16),
    dict(W=64, D=5, n_out=25, in_xyz=191, in_dir=0, flags=mp.MLP_F16),
    dict(W=64, D=5, n_out=25, in_xyz=191, in_dir=0, flags=mp.MLP_F16 | mp.MLP_F16_HEADS),
    dict(W=64, D=5, n_out=36, in_xyz=191, in_dir=0, flags=mp.MLP_F16 | mp.MLP_F16_HEADS),
    # 8 x 256: the rgb head alone is paired
    dict(W=256, D=8, n_out=3, in_xyz=63, in_dir=91, flags=mp.MLP_F16 | mp.MLP_F16_HEADS | mp.MLP_SIGMOID | mp.MLP_WITH_SIGMA),
    dict(W=256, D=8, n_out=3, in_xyz=63, in_dir=27, flags=mp.MLP_F16 | mp.MLP_F16_HEADS),
]


@pytest.mark.parametrize("kw", SPECS)
def test_packer_agrees_with_library_stream_shape(lib, kw):
    spec = mp.MlpSpec(n_freq=10, **kw)
    idx = mp.stream_index(spec)
    d = _lib.MlpDesc(W=spec.W, D=spec.D, n_out=spec.n_out, flags=spec.flags, n_freq=10, reserved=0)
    assert lib.moda_mlp_stream_bytes(ctypes.byref(d)) == idx.stream_bytes
    assert lib.moda_mlp_bias_floats(ctypes.byref(d)) == spec.nbias == idx.bidx.shape[0]
    if spec.x3:      # every real fragment twice, as (values, residuals); nothing else differs from the bf16 stream's content
        plain = mp.stream_index(mp.MlpSpec(n_freq=10, **dict(kw, flags=(kw["flags"] & ~mp.MLP_BF16X3) | mp.MLP_BF16)))
        fr = idx.widx.reshape(-1, 512)
        real = ~(fr == idx.zero).all(1)
        pfr = plain.widx.reshape(-1, 512)
        preal = ~(pfr == plain.zero).all(1)
        assert real.sum() == 2 * preal.sum() == 2 * idx.part.sum()
        assert np.array_equal(fr[real][0::2], pfr[preal]) and np.array_equal(fr[real][1::2], pfr[preal])
        assert np.array_equal(idx.part[real], np.tile([0, 1], preal.sum()))
    if spec.f16 and not spec.heads_split:      # the fp16 stream is the bf16 stream's layout with fp16 elements
        b16 = mp.stream_index(mp.MlpSpec(n_freq=10, **dict(kw, flags=(kw["flags"] & ~mp.MLP_F16) | mp.MLP_BF16)))
        assert np.array_equal(idx.widx, b16.widx) and idx.part.sum() == 0
    if spec.heads_split:                        # MLP_F16_HEADS: only the dir and rgb layers' fragments come as (values, residuals) pairs
        names = mp.weight_names(spec)
        wcode = idx.codes()[0].reshape(-1, 512)
        sid = np.where(wcode[:, 0] >= 0, (wcode[:, 0] >> 24) & 15, -1)
        heads = {names.index("rgb.0.weight")} | ({names.index("dir_encoding.0.weight")} if spec.W == 64 else set())
        lo = idx.part.astype(bool)
        assert lo.sum() == (spec.NTD * spec.NT * spec.subs if spec.W == 64 else 0) + ((spec.n_out + 31) // 32) * spec.NTD * spec.subs
        assert all(int(sid[i]) in heads for i in np.nonzero(lo)[0])
        assert all(np.array_equal(idx.widx.reshape(-1, 512)[i], idx.widx.reshape(-1, 512)[i - 1]) for i in np.nonzero(lo)[0])
        with pytest.raises(ValueError):
            mp.MlpSpec(n_freq=10, **dict(kw, flags=mp.MLP_BF16 | mp.MLP_F16_HEADS)).check()
        with pytest.raises(ValueError):          # no split heads for the sigma-only pass or the 128-wide network
            mp.MlpSpec(n_freq=10, **dict(kw, flags=mp.MLP_F16 | mp.MLP_F16_HEADS | mp.MLP_SIGMA_ONLY)).check()
        with pytest.raises(ValueError):
            mp.MlpSpec(n_freq=10, **dict(kw, W=128, flags=mp.MLP_F16 | mp.MLP_F16_HEADS)).check()


def test_unsupported_shapes_are_refused(lib):
    d = _lib.MlpDesc(W=96, D=8, n_out=3, flags=0, n_freq=10, reserved=0)
    assert lib.moda_mlp_stream_bytes(ctypes.byref(d)) == -1
    with pytest.raises(NotImplementedError):
        mp.MlpSpec(W=96, D=8, n_out=3, in_xyz=63, in_dir=0).check()
    with pytest.raises(NotImplementedError):
        moda_amd.NeRF(enable_semantic=True)


def test_no_cpu_fallback():
    """CPU tensors are refused: the HIP library is the only compute path."""
    with pytest.raises(RuntimeError, match="CUDA"):
        moda_amd.dq_inverse(torch.zeros(2, 8))
    with pytest.raises(RuntimeError, match="CUDA"):
        moda_amd.Embedding(3, 10)(torch.zeros(2, 3))
    m = moda_amd.NeRF()
    with torch.no_grad(), pytest.raises(RuntimeError, match="CUDA"):
        m(torch.zeros(2, 90))


def test_state_dict_names_match_reference():
    """Checkpoint compatibility: the names the reference's NeRF registers (nerf.py:111-140)."""
    m = moda_amd.NeRF(D=8, W=256, in_channels_xyz=63, in_channels_dir=91)
    keys = set(m.state_dict())
    want = {f"xyz_encoding_{i}.0.{p}" for i in range(1, 9) for p in ("weight", "bias")}
    want |= {f"{n}.{p}" for n in ("xyz_encoding_final", "dir_encoding.0", "sigma", "rgb.0") for p in ("weight", "bias")}
    want |= {"beta"}
    assert keys == want
    assert m.xyz_encoding_5[0].weight.shape == (256, 319) and m.dir_encoding[0].weight.shape == (128, 347)
    assert m.skips == [4] and m.weights_reg == ["xyz_encoding_1", "xyz_encoding_5"]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="missing"):
        _lib.load()


def test_initial_precision_comes_from_the_environment_and_defaults_to_the_parity_grade_fp16_mode():
    """a caller that only swaps its imports must not land on the exact-fp32 kernels (11x slower at config 2 for
    the same 1e-4 bar): the library starts in MODA_PRECISION, default 'fp16'; MODA_TRAIN_PRECISION likewise for the training route
    (default: exact fp32).  Checked in fresh interpreters (this process has the suite's MODA_PRECISION=fp32, tests/conftest.py)."""
    import subprocess
    import sys

    def probe(**env_over):
        env = {k: v for k, v in os.environ.items() if k not in ("MODA_PRECISION", "MODA_TRAIN_PRECISION")}
        env.update(env_over)
        p = subprocess.run([sys.executable, "-c", "import moda_amd; print(moda_amd.get_precision(), moda_amd.get_train_precision())"],
                           env=env, capture_output=True, text=True, cwd=ROOT, timeout=300)
        return p.returncode, p.stdout.strip(), p.stderr
    assert probe()[:2] == (0, "fp16 fp32")
    assert probe(MODA_PRECISION="bf16x3", MODA_TRAIN_PRECISION="bf16x6")[:2] == (0, "bf16x3 bf16x6")
    assert probe(MODA_PRECISION="fp32")[:2] == (0, "fp32 fp32")
    rc, _, err = probe(MODA_PRECISION="fp8")
    assert rc != 0 and "MODA_PRECISION" in err
    assert moda_amd.get_precision() == "fp32"            # this process: the suite's exact baseline
