"""8 x 256 bf16 inference kernel, interleaved A/B in ONE process on one box: the eight-wave one-column-block form (A) against the
four-wave two-column-block form with the activations in asm-owned AGPRs (B, MODA_MLP_AGPR=1; PrecBF16A in mlp_fused.hip).
Bit-equality of the outputs first (same arithmetic per sample: the forms differ in which registers hold what), then alternating
timed launches at config 2's size, then the whole render_rays step.   usage: python tools/agpr_ab.py [rounds]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

import moda_amd
from moda_amd import synth
from moda_amd.bench_support import make_models, make_opts, rays_to_gpu, nerf_from_params, T

torch.set_grad_enabled(False)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16"          # bf16 (PrecBF16A) or fp16 (PrecF16A, split rgb head)
N, S = 65536, 256
M = N * S
kw = dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False)
p = synth.nerf_params(5, "mb/coarse", D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3)
m = nerf_from_params(p, **kw)
xyz = torch.from_numpy(np.float32(0.3) * synth.normal(5, "mb/xyz", (4096 * 16, 3))).cuda().repeat(M // (4096 * 16), 1).contiguous()
dirs = T(synth.normal(5, "mb/dir", (N, 91)))


def run(agpr, prec=None):
    prec = prec or PREC
    os.environ["MODA_MLP_AGPR"] = "1" if agpr else "0"
    return m.fused(xyz.view(N, S, 3), dir_src=dirs, precision=prec)


a, b = run(False), run(True)
torch.cuda.synchronize()
same = torch.equal(a, b)
print(f"outputs bit-identical: {same}; max |diff| {float((a - b).abs().max()):.3e}; any NaN {bool(torch.isnan(b).any())}")
if not same:      # where do they differ?  sample m of a 256-sample workgroup tile: wave (m % 256) // 64, column block ((m % 256) // 32) % 2
    d = (a.reshape(-1, 4) != b.reshape(-1, 4))
    bad = d.any(1).nonzero().reshape(-1)
    print(f"   {bad.numel()} of {M} samples differ ({bad.numel() / M:.2%}); per channel {d.sum(0).tolist()}")
    pos = (bad % 256).cpu().numpy()
    print("   by wave:", np.bincount(pos // 64, minlength=4).tolist(), " by column block:", np.bincount((pos // 32) % 2, minlength=2).tolist(),
          " by lane column (first 8):", np.bincount(pos % 32, minlength=32)[:8].tolist())
    tiles = (bad // 256).cpu().numpy()
    print("   tiles affected:", len(np.unique(tiles)), "of", M // 256, " first few:", np.unique(tiles)[:10].tolist(),
          " mean |diff| on bad samples", float((a.reshape(-1, 4)[bad] - b.reshape(-1, 4)[bad]).abs().mean()))
    b2 = run(True)
    print("   B repeatable:", torch.equal(b, b2))
# ragged sizes and short batches
for n_, s_ in ((7, 32), (513, 64), (4096, 96), (1000, 256)):
    x2 = xyz[:n_ * s_].view(n_, s_, 3); d2 = dirs[:n_]
    os.environ["MODA_MLP_AGPR"] = "0"; r0 = m.fused(x2, dir_src=d2, precision=PREC)
    os.environ["MODA_MLP_AGPR"] = "1"; r1 = m.fused(x2, dir_src=d2, precision=PREC)
    print(f"   {n_} x {s_}: identical {torch.equal(r0, r1)}")
    same = same and torch.equal(r0, r1)


def timed(agpr, n=4):
    ts = []
    for _ in range(n):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); run(agpr); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return ts


for _ in range(3):
    run(False); run(True)
torch.cuda.synchronize()
ta, tb = [], []
for r in range(rounds):
    ta += timed(False); tb += timed(True)
fl = 2 * 601600 * M
print(f"A (8 waves x 1 block): median {np.median(ta):.3f} ms min {np.min(ta):.3f}  = {fl / np.median(ta) / 1e9:.0f} TFLOP/s algorithmic")
print(f"B (4 waves x 2 blocks, AGPR): median {np.median(tb):.3f} ms min {np.min(tb):.3f}  = {fl / np.median(tb) / 1e9:.0f} TFLOP/s algorithmic")
print(f"B / A = {np.median(tb) / np.median(ta):.4f}")
# whole step
models, emb = make_models(0, 25)
rays = rays_to_gpu(synth.make_rays(1000, N, 25, rays_per_frame=256))
moda_amd.set_precision(PREC)
opts = make_opts()


def step(agpr):
    os.environ["MODA_MLP_AGPR"] = "1" if agpr else "0"
    return moda_amd.render_rays(models, emb, rays, N_samples=S, perturb=0, noise_std=0.0, opts=opts, img_size=512)


ra, rb = step(False), step(True)
torch.cuda.synchronize()
print("render_rays outputs identical:", all(torch.equal(ra[k], rb[k]) for k in ("img_coarse", "depth_rnd", "sil_coarse")))
sa, sb = [], []
for r in range(rounds):
    for ag, dst in ((False, sa), (True, sb)):
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            step(ag)
        e.record(); torch.cuda.synchronize(); dst.append(s.elapsed_time(e) / 5)
print(f"render_rays step: A {np.median(sa):.3f} ms ({N / np.median(sa) / 1e3:.3f} M rays/s), B {np.median(sb):.3f} ms ({N / np.median(sb) / 1e3:.3f} M rays/s), B / A = {np.median(sb) / np.median(sa):.4f}")
os.environ["MODA_MLP_AGPR"] = "0"
sys.exit(0 if same else 1)
