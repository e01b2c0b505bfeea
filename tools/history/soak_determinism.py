"""Soak: the forward path has no atomics, so repeated launches on the same inputs must be bit-identical; a race in the
ring / LDS staging of the fused MLP kernel would show up as a differing checksum.  usage: soak_determinism.py [steps]"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import moda_amd
from moda_amd import synth
from gpu_helpers import make_models, make_opts, rays_to_gpu
torch.set_grad_enabled(False)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for prec, N, S in (("bf16", 65536, 256), ("fp32", 4096, 128), ("bf16", 4099, 50)):
    moda_amd.set_precision(prec)
    models, emb = make_models(3, 25, with_feat=True, with_vis=True)
    rays = rays_to_gpu(synth.make_rays(3, N, 25, rays_per_frame=1 if N % 256 else 256))
    ref = None
    n = steps if prec == "bf16" and N == 65536 else max(10, steps // 5)
    for i in range(n):
        res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512,
                                   render_vis=True, obj_bound=[0.3, 0.3, 0.3])
        sig = tuple(float(res[k].double().sum()) for k in ("img_coarse", "depth_rnd", "frame_cyc_dis", "xyz_canonical_vis", "vis_pred"))
        if ref is None:
            ref = sig
            keep = {k: res[k].clone() for k in ("img_coarse", "xyz_canonical_vis")}
        assert sig == ref, (prec, N, S, i, sig, ref)
        assert torch.equal(res["img_coarse"], keep["img_coarse"]) and torch.equal(res["xyz_canonical_vis"], keep["xyz_canonical_vis"])
    print(f"{prec} {N}x{S}: {n} launches bit-identical")
moda_amd.set_precision("fp32")
