"""debug: split-bf16 vs exact-fp32 training GEMMs through a whole network -- dense small differences (precision) or sparse large ones (ReLU switches)?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
MODE = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
import moda_amd
from moda_amd import synth
from gpu_helpers import T, nerf_from_params
CASES = {"coarse_sigma": (dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False), 0, 91, True),
         "coarse": (dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False), 0, 91, False),
         "skin": (dict(D=5, W=64, in_channels_xyz=191, in_channels_dir=0, out_channels=25, raw_feat=True), 128, 0, False)}
for name, (kw, code_c, dir_c, sigma_only) in CASES.items():
    for R, S in ((7, 37), (64, 64)):
        pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
        p = synth.nerf_params(51, "nf/" + name, **pk)
        xyz = np.float32(0.3) * synth.normal(51, "nf/xyz", (R, S, 3))
        code = synth.normal(51, "nf/code", (R, code_c)) if code_c else None
        dirs = synth.normal(51, "nf/dir", (R, dir_c)) if dir_c else None
        n_out = 1 if sigma_only else kw["out_channels"] + (0 if kw["raw_feat"] else 1)
        gout = synth.normal(51, "nf/g", (R, S, n_out))
        res = {}
        for prec in ("fp32", MODE):
            moda_amd.set_train_precision(prec)
            m = nerf_from_params(p, **kw).train()
            emb = moda_amd.Embedding(3, 10)
            xg = T(xyz).requires_grad_(True)
            cg = None if code is None else T(code).requires_grad_(True)
            dg = None if dirs is None else T(dirs).requires_grad_(True)
            yg = m.train_forward(xg, emb, code=cg, dir_src=dg, sigma_only=sigma_only)
            (yg * T(gout)).sum().backward()
            res[prec] = (yg.detach().double(), xg.grad.double(), {n: q.grad.double() for n, q in m.named_parameters() if q.grad is not None})
        moda_amd.set_train_precision("fp32")
        y0, g0, p0 = res["fp32"]; y1, g1, p1 = res[MODE]
        e = (g1 - g0).abs().amax(-1).flatten() / g0.abs().max()
        print(f"{name} {R}x{S}: y max {float((y1 - y0).abs().max() / y0.abs().max()):.1e} | dxyz max {float(e.max()):.1e} "
              f"L2 {float((g1 - g0).norm() / g0.norm()):.1e} samples>1e-5: {int((e > 1e-5).sum())}/{e.numel()} median {float(e.median()):.1e} | "
              f"params worst L2 {max(float((p1[n] - p0[n]).norm() / (p0[n].norm() + 1e-30)) for n in p0):.1e}")
