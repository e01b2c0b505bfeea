"""Diagnostic: repeated launches of moda_mlp_warp_fwd on identical inputs -- where do results differ?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import moda_amd
from moda_amd import synth, geom_utils as G
from gpu_helpers import make_models, T

N, S, B = 8192, 256, 25
models, emb = make_models(0, B)
rays = synth.make_rays(1000, N, B, rays_per_frame=256)
xyz = T(rays["rays_o"][:, None] + rays["rays_d"][:, None] * np.linspace(0.1, 0.5, S, dtype=np.float32)[None, :, None])
rts = T(rays["bone_rts"]); code = T(rays["time_embedded"])
skin = models["nerf_skin"]
with torch.no_grad():
    bd = G.bone_transform(models["bones_rst"], rts, True, is_vec=True)
    outs = []
    for i in range(4):
        o, _ = skin.fused_warp(xyz, emb["xyz"], code, bd, rts, models["skin_aux"], backward=True)
        outs.append(o.clone())
    torch.cuda.synchronize()
    dskin = skin.fused(xyz, n_freq=10, alpha=10.0, code=code, out_tr_S=S, precision="bf16")
    want, _, _ = G.warp(bd, rts, xyz, dskin, models["skin_aux"], backward=True, dskin_bns=True)
for i in range(1, 4):
    d = (outs[i] - outs[0]).abs().amax(-1).reshape(-1)
    idx = torch.nonzero(d > 0).reshape(-1).cpu().numpy()
    print(f"run {i} vs 0: {len(idx)} samples differ, max {d.max().item():.2e}")
    if len(idx):
        print("  sample%32 hist:", np.bincount(idx % 32, minlength=32).tolist())
        tiles = np.unique(idx // 32)
        print("  tiles touched:", len(tiles), " samples per touched tile:", len(idx) / len(tiles))
        print("  tile%16 (wave) hist:", np.bincount(tiles % 16, minlength=16).tolist())
        print("  first idx:", idx[:20].tolist())
for i in range(4):
    e = (outs[i] - want).abs().amax(-1).reshape(-1)
    print(f"run {i} vs two-kernel route: max {e.max().item():.2e}, >1e-5: {int((e > 1e-5).sum())}, >1e-4: {int((e > 1e-4).sum())}")
