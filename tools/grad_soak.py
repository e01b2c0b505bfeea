"""Repeat the cfg4 forward + backward on fixed inputs and report every gradient tensor that deviates from the median run by
more than atomics noise: python tools/grad_soak.py [bf16|fp32] [iters] [N] [S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_helpers import TrainHarness

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
S = int(sys.argv[4]) if len(sys.argv) > 4 else 128
h = TrainHarness(N=N, S=S, precision=prec, lr=5e-4, bucket=os.environ.get("BUCKET", "1") == "1")
for _ in range(2):
    h.eager_step()
h.draw()
names = []
for mk, m in h.models.items():
    if isinstance(m, torch.nn.Module):
        names += [f"{mk}.{n}" for n, _ in m.named_parameters()]
names += ["bones_rst", "skin_aux"]
leaf_keys = [k for k, v in h.rays.items() if torch.is_tensor(v) and v.requires_grad]
runs, losses = [], []
# LOAD=1: an unrelated MFMA-heavy stream runs beside every evaluation (do kernels that share CUs with other work still compute
# the same thing?)
bg = None
if os.environ.get("LOAD") == "1":
    bg = torch.cuda.Stream()
    xa = torch.randn(4096, 4096, device=h.dev, dtype=torch.bfloat16)
    xb = torch.randn(4096, 4096, device=h.dev, dtype=torch.bfloat16)
for it in range(iters):
    if bg is not None:
        with torch.cuda.stream(bg):
            for _ in range(40):
                xc = xa @ xb
    h.zero_grad()
    for k in leaf_keys:
        h.rays[k].grad = None
    losses.append(float(h.fwd_bwd()))
    g = [None if p.grad is None else p.grad.detach().clone() for p in h.params] + [None if h.rays[k].grad is None else h.rays[k].grad.detach().clone() for k in leaf_keys]
    runs.append(g)
allnames = names + ["rays." + k for k in leaf_keys]
bad = 0
for j, nm in enumerate(allnames):
    gs = [r[j] for r in runs]
    if gs[0] is None:
        continue
    st = torch.stack(gs).double()
    med = st.median(0).values
    nrm = float(med.norm()) or 1.0
    dev = [(float((st[i] - med).norm()) / nrm) for i in range(iters)]
    base = sorted(dev)[iters // 2]
    out = [(i, d) for i, d in enumerate(dev) if d > max(20 * base, 1e-4)]
    if out:
        bad += len(out)
        i, d = max(out, key=lambda t: t[1])
        diff = (st[i] - med).abs()
        flat = diff.flatten()
        top = torch.topk(flat, min(5, flat.numel()))
        print(f"{nm}: {len(out)} of {iters} runs deviate (typical {base:.1e}); worst run {i}: rel-L2 {d:.2e}; shape {tuple(med.shape)}; "
              f"largest |diff| at flat indices {top.indices.tolist()} = {[f'{v:.3g}' for v in top.values.tolist()]}, #elements off by > 1e-3 of max: "
              f"{int((diff > 1e-3 * med.abs().max()).sum())}")
print(f"{prec}: {iters} runs, loss min/max {min(losses):.7f}/{max(losses):.7f}, deviating (tensor, run) pairs: {bad}")
