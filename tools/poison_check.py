"""Read-before-write detector: run the cfg4 forward + backward (and a no-grad render) twice -- once with every torch.empty /
empty_like buffer pre-filled with zeros, once pre-filled with garbage (NaN for floats) -- and compare every output and gradient.
A kernel that reads a buffer element it (or an earlier kernel) never wrote shows up as a difference.
usage: python tools/poison_check.py [bf16|fp32] [N] [S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import moda_amd
from gpu_helpers import TrainHarness, make_opts

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
MODE = {"fill": None, "lds": None}
_empty, _empty_like = torch.empty, torch.empty_like


def _poison(t):
    if MODE["fill"] is None or not t.is_cuda:
        return t
    if t.dtype.is_floating_point:
        t.fill_(float("nan") if MODE["fill"] == "nan" else 0.0)
    else:
        t.fill_(0 if MODE["fill"] == "zero" else (-1 if t.dtype != torch.uint8 else 255))
    return t


from moda_amd import _lib as _L
_call = _L.call


def _call_poisoned(name, *a):          # LDS of every CU overwritten before each library launch
    if MODE["lds"] is not None and name != "moda_dbg_poison_lds":
        _call("moda_dbg_poison_lds", MODE["lds"], _L.stream())
    return _call(name, *a)


_L.call = _call_poisoned
torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))

h = TrainHarness(N=N, S=S, precision=prec, lr=5e-4, bucket=os.environ.get("BUCKET", "1") == "1")
for _ in range(2):
    h.eager_step()
h.draw()
leaf_keys = [k for k, v in h.rays.items() if torch.is_tensor(v) and v.requires_grad]
names = []
for mk, m in h.models.items():
    if isinstance(m, torch.nn.Module):
        names += [f"{mk}.{n}" for n, _ in m.named_parameters()]
names += ["bones_rst", "skin_aux"] + ["rays." + k for k in leaf_keys]


def run(fill, lds=None):
    MODE["fill"] = fill
    MODE["lds"] = lds
    h.zero_grad()
    for k in leaf_keys:
        h.rays[k].grad = None
    loss = float(h.fwd_bwd())
    g = [None if p.grad is None else p.grad.detach().clone() for p in h.params]
    g += [None if h.rays[k].grad is None else h.rays[k].grad.detach().clone() for k in leaf_keys]
    MODE["fill"] = None
    MODE["lds"] = None
    return loss, g, h.terms.clone()


l0, g0, t0 = run("zero")
l1, g1, t1 = run("nan")
print(f"{prec}: loss zero-filled {l0:.7f}, NaN-filled {l1:.7f}; terms equal: {bool(torch.allclose(t0, t1, rtol=1e-5, equal_nan=False))}")
bad = 0
for nm, a, b in zip(names, g0, g1):
    if a is None:
        continue
    if not torch.isfinite(b).all() or float((a - b).norm()) > 1e-3 * max(float(a.norm()), 1e-30):
        bad += 1
        print(f"  {nm}: zero-filled vs NaN-filled differ: non-finite {int((~torch.isfinite(b)).sum())} of {b.numel()}, "
              f"rel-L2 {float((a - torch.nan_to_num(b)).norm() / max(float(a.norm()), 1e-30)):.2e}")
print("gradient tensors that depend on uninitialised memory:", bad)
# LDS left behind by earlier kernels: zeros vs NaN patterns vs a large finite pattern
la, ga, _ = run("zero", 0x00000000)
for pat in (0x7fc00000, 0x7f7fffff, 0x3f800000):
    lb, gb, _ = run("zero", pat)
    badl = 0
    for nm, a, b in zip(names, ga, gb):
        if a is None:
            continue
        if not torch.isfinite(b).all() or float((a - b).norm()) > 1e-3 * max(float(a.norm()), 1e-30):
            badl += 1
            print(f"  LDS pattern {pat:#x}: {nm} differs: non-finite {int((~torch.isfinite(b)).sum())}, rel-L2 "
                  f"{float((a - torch.nan_to_num(b)).norm() / max(float(a.norm()), 1e-30)):.2e}")
    print(f"LDS pattern {pat:#x}: loss {la:.7f} vs {lb:.7f}; gradient tensors that depend on stale LDS: {badl}")
# inference route
rays = {k: v.detach() for k, v in h.rays.items() if k in ("rays_o", "rays_d", "near", "far", "xys", "time_embedded", "bone_rts", "env_code")}
for mode in ("bf16", "bf16x3", "fp32", "fp16"):
    moda_amd.set_precision(mode)
    outs = []
    for fill in ("zero", "nan"):
        MODE["fill"] = fill
        with torch.no_grad():
            for m in h.models.values():
                if isinstance(m, torch.nn.Module):
                    m.eval()
            r = moda_amd.render_rays(h.models, h.emb, rays, N_samples=S, perturb=0, noise_std=0.0, opts=make_opts(), img_size=512)
        MODE["fill"] = None
        outs.append({k: v.clone() for k, v in r.items() if torch.is_tensor(v)})
    diff = [k for k in outs[0] if not torch.equal(torch.nan_to_num(outs[0][k].float(), nan=-7.0), torch.nan_to_num(outs[1][k].float(), nan=-7.0))]
    print(f"render ({mode}): result keys that depend on uninitialised memory: {diff}")
moda_amd.set_precision("fp32")
