"""bf16-mode training backward of the 8 x 256 network at cfg4's size (2048 rays x 128 samples), interleaved A/B in one process:
gemm_bf16.hip's dW form + dX form per hidden layer (A, MODA_BWD256=0) against bwd256_fused.hip's one launch per layer (B).
Gradient agreement first, then alternating timed forward + backward passes.   usage: python tools/bwd256_ab.py [rounds] [rays] [samples]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

import moda_amd
from moda_amd import synth
from moda_amd.bench_support import nerf_from_params, T

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
S = int(sys.argv[3]) if len(sys.argv) > 3 else 128
kw = dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False)
p = synth.nerf_params(5, "ab/coarse", D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3)
m = nerf_from_params(p, **kw).train()
emb = moda_amd.Embedding(3, 10)
xyz = T(np.float32(0.3) * synth.normal(5, "ab/xyz", (R, S, 3)))
dirs = T(synth.normal(5, "ab/dir", (R, 91)))
gout = T(synth.normal(5, "ab/g", (R, S, 4)))
moda_amd.set_train_precision("bf16")


def step(fused):
    os.environ["MODA_BWD256"] = "1" if fused else "0"
    for q in m.parameters():
        q.grad = None
    xg = xyz.clone().requires_grad_(True)
    out = m.train_forward(xg, emb, dir_src=dirs)
    (out * gout).sum().backward()
    return xg.grad, [q.grad for q in m.parameters() if q.grad is not None]


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


xa, ga = step(False)
xb, gb = step(True)
torch.cuda.synchronize()
print(f"d_xyz rel-L2 {rel(xb, xa):.3e} (max |diff| {float((xa - xb).abs().max()):.3e}); worst parameter gradient rel-L2 "
      f"{max(rel(b, a) for a, b in zip(ga, gb)):.3e}")


def timed(fused, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); step(fused); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return ts


for _ in range(2):
    step(False); step(True)
ta, tb = [], []
for r in range(rounds):
    ta += timed(False); tb += timed(True)
print(f"forward + backward, {R} x {S} samples: A (two launches per layer) median {np.median(ta):.3f} ms, B (fused) median {np.median(tb):.3f} ms, "
      f"B - A = {np.median(tb) - np.median(ta):+.3f} ms")
