"""LinearFn in the bf16x6 precision against float64 over output widths that are multiples of 4 but not of 8 (the skin head at 36 bones)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import moda_amd
from moda_amd import synth, autograd as A
from helpers import rel_l2
from gpu_helpers import T

for prec in ("bf16x6", "bf16x3", "fp32"):
    moda_amd.set_train_precision(prec)
    for M in (768, 2048):
        for K, O, act in ((32, 36, 0), (32, 40, 0), (32, 25, 0), (32, 44, 0), (64, 36, 1), (36, 32, 1), (32, 36, 1), (256, 36, 0), (32, 100, 0)):
            x = synth.normal(31, "lp/x", (M, K)); W = synth.normal(31, "lp/w", (O, K)) * np.float32(0.1); b = synth.normal(31, "lp/b", (O,))
            g = synth.normal(31, "lp/g", (M, O))
            xc, Wc, bc = (torch.from_numpy(a).double().requires_grad_(True) for a in (x, W, b))
            z = xc @ Wc.T + bc
            yc = torch.relu(z) if act == 1 else z
            (yc * torch.from_numpy(g).double()).sum().backward()
            xg, Wg, bg = (T(a).requires_grad_(True) for a in (x, W, b))
            yg = A.LinearFn.apply(xg, Wg, bg, act)
            (yg * T(g)).sum().backward()
            e = [rel_l2(yg.detach().cpu().numpy(), yc.detach().numpy()), rel_l2(xg.grad.cpu().numpy(), xc.grad.numpy()),
                 rel_l2(Wg.grad.cpu().numpy(), Wc.grad.numpy()), rel_l2(bg.grad.cpu().numpy(), bc.grad.numpy())]
            flag = "  <<<<" if max(e) > 1e-5 and prec != "bf16x3" else ""
            print(f"{prec:6s} M {M:5d} K {K:3d} O {O:3d} act {act}: y {e[0]:.1e} dx {e[1]:.1e} dW {e[2]:.1e} db {e[3]:.1e}{flag}", flush=True)
moda_amd.set_train_precision("fp32")
