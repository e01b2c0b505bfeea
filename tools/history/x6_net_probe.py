"""NerfFn (whole skin network) in bf16x6 / fp32 against float64, over n_out and M: which shape loses accuracy?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import moda_amd
from moda_amd import synth
from oracle import torch_ref as tr
from helpers import rel_l2
from gpu_helpers import T, nerf_from_params

TC = torch.from_numpy
emb = moda_amd.Embedding(3, 10)
for n_out in (36, 25, 40):
    for R, S in ((64, 12), (64, 32), (16, 128), (1, 2048), (64, 31)):
        kw = dict(D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=n_out, raw_feat=True)
        pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
        p = synth.nerf_params(53, "np/skin", **pk)
        xyz = np.float32(0.3) * synth.normal(53, "np/xyz", (R, S, 3))
        code = synth.normal(53, "np/code", (R, 128))
        gout = synth.normal(53, "np/g", (R, S, n_out))
        pc = {k: TC(v).double().requires_grad_(True) for k, v in p.items()}
        xc = TC(xyz).double().requires_grad_(True)
        cc = TC(code).double().requires_grad_(True)
        cols = [tr.embedding(xc, 10, 10.0), cc[:, None].expand(R, S, 128)]
        yc = tr.nerf_forward(pc, torch.cat(cols, -1), 5, 64, 191, 0, raw_feat=True)
        (yc * TC(gout).double()).sum().backward()
        for prec in ("fp32", "bf16x6"):
            moda_amd.set_train_precision(prec)
            m = nerf_from_params(p, **kw).train()
            xg = T(xyz).requires_grad_(True)
            cg = T(code).requires_grad_(True)
            yg = m.train_forward(xg, emb, code=cg)
            (yg * T(gout)).sum().backward()
            moda_amd.set_train_precision("fp32")
            errs = [("y", rel_l2(yg.detach().cpu().numpy(), yc.detach().numpy())), ("d_xyz", rel_l2(xg.grad.cpu().numpy(), xc.grad.numpy())),
                    ("d_code", rel_l2(cg.grad.cpu().numpy(), cc.grad.numpy()))]
            for pn, pt in m.named_parameters():
                if pt.grad is not None and pc[pn].grad is not None and float(pc[pn].grad.abs().max()) > 0:
                    errs.append((pn, rel_l2(pt.grad.cpu().numpy(), pc[pn].grad.numpy())))
            errs.sort(key=lambda e: -e[1])
            print(f"n_out {n_out} R {R} S {S} M {R * S} [{prec:6s}] y {dict(errs)['y']:.1e} worst: " + ", ".join(f"{k}={e:.1e}" for k, e in errs[:4]), flush=True)
