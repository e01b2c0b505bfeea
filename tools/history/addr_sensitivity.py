"""Diagnostic: does the fused warp kernel's time depend on where its tensors land in memory?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import moda_amd
from moda_amd import synth, geom_utils as G
from gpu_helpers import make_models, T
N, S, B = 65536, 256, 25
models, emb = make_models(0, B)
rays = synth.make_rays(1000, N, B, rays_per_frame=256)
rts = T(rays["bone_rts"]); code = T(rays["time_embedded"])
skin = models["nerf_skin"]
keep = []
def run(label, shift_bytes):
    if shift_bytes:
        keep.append(torch.empty(shift_bytes, dtype=torch.uint8, device="cuda"))
    xyz = torch.empty((N, S, 3), device="cuda").uniform_(-0.3, 0.3)
    with torch.no_grad():
        bd = G.bone_transform(models["bones_rst"], rts, True, is_vec=True)
        for _ in range(3):
            o, _ = skin.fused_warp(xyz, emb["xyz"], code, bd, rts, models["skin_aux"], backward=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); o, _ = skin.fused_warp(xyz, emb["xyz"], code, bd, rts, models["skin_aux"], backward=True); e.record()
            torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    print(f"{label}: {np.median(ts):.3f} ms (min {min(ts):.3f}); xyz @ {xyz.data_ptr():#x} out @ {o.data_ptr():#x} delta {(o.data_ptr()-xyz.data_ptr())/2**20:.3f} MiB")
    del xyz, o
for lab, sh in (("base", 0), ("+4KB", 4096), ("+68KB", 64 * 1024), ("+1MB+8K", 2**20 + 8192), ("+33MB", 33 * 2**20), ("+7MB", 7 * 2**20), ("+2MB", 2 * 2**20)):
    run(lab, sh)
