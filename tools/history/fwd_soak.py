"""Long bit-identity soak of every fused inference kernel variety (one-off; the suite's soak tests run the short form):
python tools/fwd_soak.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import moda_amd
from moda_amd import synth
from gpu_helpers import T, make_models, rays_to_gpu
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
torch.set_grad_enabled(False)
models, emb = make_models(5, 25, with_feat=True, with_vis=True)
N, S = 1031, 128                                    # ragged: 131 968 samples + a partial workgroup tile
xyz = T(np.float32(0.3) * synth.normal(5, "fs/xyz", (N, S, 3)))
dirs = T(synth.normal(5, "fs/dir", (N, 91)))
code = T(synth.normal(5, "fs/code", (N, 128)))
rays = rays_to_gpu(synth.make_rays(5, N, 25, rays_per_frame=1))
bones = moda_amd.bone_transform(models["bones_rst"], rays["bone_rts"], True, is_vec=True)
bg = torch.cuda.Stream()
xa = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16); xb = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
cases = []
for prec in ("bf16", "fp16", "bf16x3", "fp32"):
    r = reps if prec != "fp32" else max(reps // 8, 50)
    cases += [(f"coarse {prec}", r, lambda p=prec: models["coarse"].fused(xyz, dir_src=dirs, precision=p)),
              (f"coarse sigma_only {prec}", r, lambda p=prec: models["coarse"].fused(xyz, sigma_only=True, precision=p)),
              (f"feat 5x128 {prec}", r, lambda p=prec: models["nerf_feat"].fused(xyz, precision=p)),
              (f"skin 5x64 {prec}", r, lambda p=prec: models["nerf_skin"].fused(xyz, code=code, precision=p, out_tr_S=S))]
for prec in ("bf16", "fp16", "bf16x3"):
    cases.append((f"skin+warp {prec}", reps, lambda p=prec: models["nerf_skin"].fused_warp(xyz, emb["xyz"], code, bones, rays["bone_rts"], models["skin_aux"],
                                                                                       backward=True, precision=p)[0]))
total_bad = 0
for name, r, fn in cases:
    ref = fn().clone()
    bad = 0
    for it in range(r):
        if it % 8 == 0:
            with torch.cuda.stream(bg):
                for _ in range(4):
                    xc = xa @ xb
        if not torch.equal(fn(), ref):
            bad += 1
    total_bad += bad
    print(f"{name:28s} {bad} of {r} launches differ", flush=True)
print("total differing launches:", total_bad)
