"""Repeat the training forward (moda_mlp_dump_fwd) of one network on fixed inputs beside an MFMA load and report every launch whose
output differs from the first: python tools/dump_fwd_repro.py [feat|skin|coarse|vis] [M] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import moda_amd
from moda_amd import synth
from gpu_helpers import T, make_models
net = sys.argv[1] if len(sys.argv) > 1 else "feat"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 270144
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
moda_amd.set_train_precision("bf16")
models, emb = make_models(6, 25, with_feat=True, with_vis=True)
m = {"feat": models["nerf_feat"], "skin": models["nerf_skin"], "coarse": models["coarse"], "vis": models["nerf_vis"]}[net].train()
xyz = T(np.float32(0.3) * synth.normal(6, "rep/xyz", (M, 3)))
kw = {}
if net == "skin":
    kw["code"] = T(synth.normal(6, "rep/code", (1, 128)))
if net == "coarse":
    kw["dir_src"] = T(synth.normal(6, "rep/dir", (1, 91)))
bg = torch.cuda.Stream()
xa = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
xb = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
load = os.environ.get("LOAD", "1") == "1"
with torch.enable_grad():
    ref = m.train_forward(xyz, emb["xyz"], **kw).detach().clone()
    bad = 0
    for it in range(reps):
        if load and it % 8 == 0:
            with torch.cuda.stream(bg):
                for _ in range(6):
                    xc = xa @ xb
        out = m.train_forward(xyz, emb["xyz"], **kw).detach()
        if not torch.equal(out, ref):
            bad += 1
            d = (out - ref).abs()
            rows = torch.nonzero(d.amax(1) > 0).flatten()
            print(f"launch {it}: {rows.numel()} rows differ, rows {int(rows.min())}..{int(rows.max())} (tile {int(rows.min()) // 128}..), max |diff| {float(d.max()):.2e}, "
                  f"ref max {float(ref.abs().max()):.2f}", flush=True)
print(f"{net} M={M}: {bad} of {reps} launches differ (LOAD={int(load)})")
