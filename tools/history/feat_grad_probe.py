"""Localise the rare outlier in nerf_feat's gradients (tools/grad_soak.py LOAD=1: one evaluation in ~1000 has every nerf_feat
gradient 2-3e-3 off): record, per evaluation, the gradients that ENTER the feature network -- d(lattice features) from the matching
head and d(rendered features) from compositing -- and the matching head's own inputs / outputs, and report which of them deviates
from the median evaluation when a parameter gradient does.   usage: LOAD=1 python tools/feat_grad_probe.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import moda_amd
from moda_amd import autograd as A, loss_utils as LU
from gpu_helpers import TrainHarness

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
h = TrainHarness(N=2048, S=128, precision="bf16", lr=5e-4)
for _ in range(2):
    h.eager_step()
h.draw()
rec = {}


def keep(name):
    def hook(g):
        rec[name] = g.detach().clone()
        return None
    return hook


orig_fm = A.FeatMatchFn.apply


def fm(fn, vn, query, kappa, use_ot, want_prob=False):
    if fn.requires_grad:
        fn.register_hook(keep("d_feats_n"))
    if vn.requires_grad:
        vn.register_hook(keep("d_vol_n (lattice, normalised)"))
    rec["in_vol_n"] = vn.detach().clone()
    out = orig_fm(fn, vn, query, kappa, use_ot, want_prob)
    rec["out_pts_pred"] = out[0].detach().clone()
    out[0].register_hook(keep("d_pts_pred"))
    return out


A.FeatMatchFn.apply = staticmethod(fm) if False else fm
orig_nerf = A.NerfFn.apply


def nerf_apply(spec, xyz, code, dir_src, *params):
    out = orig_nerf(spec, xyz, code, dir_src, *params)
    if spec.W == 128 and out.requires_grad:
        out.register_hook(keep("d_nerf_feat_output (rendered + lattice rows)"))
        rec["nerf_feat_output"] = out.detach().clone()
    return out


A.NerfFn.apply = nerf_apply
bg = None
if os.environ.get("LOAD") == "1":
    bg = torch.cuda.Stream()
    xa = torch.randn(4096, 4096, device=h.dev, dtype=torch.bfloat16)
    xb = torch.randn(4096, 4096, device=h.dev, dtype=torch.bfloat16)
feat_params = [(n, p) for n, p in h.models["nerf_feat"].named_parameters()]
runs = []
for it in range(iters):
    if bg is not None:
        with torch.cuda.stream(bg):
            for _ in range(40):
                xc = xa @ xb
    rec.clear()
    h.zero_grad()
    for k, v in h.rays.items():
        if torch.is_tensor(v) and v.requires_grad:
            v.grad = None
    h.fwd_bwd()
    snap = dict(rec)
    for n, p in feat_params:
        if p.grad is not None and n in ("rgb.0.weight", "xyz_encoding_3.0.weight"):
            snap["GRAD " + n] = p.grad.detach().clone()
    runs.append(snap)
keys = sorted(runs[0])
print("recorded:", keys)
med = {k: torch.stack([r[k] for r in runs[:31]]).double().median(0).values for k in keys}
dev = {k: [float((r[k].double() - med[k]).norm() / max(float(med[k].norm()), 1e-30)) for r in runs] for k in keys}
typ = {k: sorted(v)[len(v) // 2] for k, v in dev.items()}
print("typical deviation:", {k: f"{v:.1e}" for k, v in typ.items()})
hits = [i for i in range(iters) if any(dev[k][i] > max(50 * typ[k], 3e-4) for k in keys if k.startswith("GRAD"))]
print("outlier evaluations:", hits)
for i in hits[:6]:
    print(f"evaluation {i}:", {k: f"{dev[k][i]:.2e}" for k in keys})
    for k in keys:
        if dev[k][i] > max(50 * typ[k], 1e-5) and not k.startswith("GRAD"):
            d = (runs[i][k].double() - med[k]).abs()
            flat = d.flatten()
            top = torch.topk(flat, min(6, flat.numel()))
            nz = int((d > 1e-3 * med[k].abs().max()).sum())
            shape = tuple(med[k].shape)
            print(f"   {k} {shape}: {nz} elements off by > 1e-3 of max; largest at flat indices {top.indices.tolist()} "
                  f"(rows {[int(j) // shape[-1] for j in top.indices.tolist()]}) = {[f'{v:.2e}' for v in top.values.tolist()]}")
