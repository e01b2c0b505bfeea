"""GPU (-m gpu): what a caller gets who ONLY swaps the imports (nnutils.rendering -> moda_amd) and sets nothing: the library's
default mode (the parity-grade fp16 mode, moda_amd/nerf.py _initial_precision) on a fresh interpreter, against the numpy oracle at
the north star's bar (1e-4 relative + the per-element figure), through render_rays and through the standalone entry points the
reference's callers use (evaluate_mlp, NeRF.forward)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
import moda_amd
from moda_amd import synth
from moda_amd.bench_support import make_models, make_opts, rays_to_gpu
from oracle import moda_oracle as orc
from helpers import oracle_scene, rel_err, elem_err
assert moda_amd.get_precision() == "fp16", moda_amd.get_precision()
N, S, B = 96, 64, 25
models, emb = make_models(11, B)
rays_np = synth.make_rays(11, N, B, rays_per_frame=16)
with torch.no_grad():
    res = moda_amd.render_rays(models, emb, rays_to_gpu(rays_np), N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
moda_amd.overflow.check()
ref = orc.render_rays(oracle_scene(11, B), rays_np, N_samples=S)
for k in ("img_coarse", "depth_rnd", "sil_coarse", "xyz_canonical_vis", "frame_cyc_dis"):
    a = res[k].cpu().numpy()
    assert rel_err(a, ref[k]) < 1e-4 and elem_err(a, ref[k]) < 1, (k, rel_err(a, ref[k]), elem_err(a, ref[k]))
# raw network outputs (no sigmoid / compositing behind them) stay on the split-bf16 kernels in this mode: ~1e-6
x = rays_to_gpu({"x": (np.float32(0.3) * synth.normal(11, "dm/x", (257, 3)))})["x"]
with torch.no_grad():
    raw = models["nerf_skin"].fused(x, code=torch.zeros(1, 128, device=x.device))
    moda_amd.set_precision("fp32")
    exact = models["nerf_skin"].fused(x, code=torch.zeros(1, 128, device=x.device))
assert float((raw - exact).abs().max() / exact.abs().max()) < 2e-5
print("default-mode ok")
'''


def test_import_swap_only_caller_renders_within_the_bar():
    env = {k: v for k, v in os.environ.items() if k not in ("MODA_PRECISION", "MODA_TRAIN_PRECISION")}
    p = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert p.returncode == 0 and "default-mode ok" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
