"""Do kernels ever write into gradient-bucket slices they do not own?  The sigma heads of the raw_feat networks (nerf_skin, nerf_feat,
nerf_vis) are never evaluated, so their bucket slices must stay exactly zero after every forward + backward.
usage: python tools/bucket_leak_probe.py [iters] [bf16|fp32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from gpu_helpers import TrainHarness
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
h = TrainHarness(N=2048, S=128, precision=prec, lr=5e-4)
for _ in range(2):
    h.eager_step()
h.draw()
b = h.bucket
names = {}
for mk, m in h.models.items():
    if isinstance(m, torch.nn.Module):
        for n, p in m.named_parameters():
            names[id(p)] = f"{mk}.{n}"
order = [(names.get(id(p), "?"), off, p.numel()) for p, off in zip(b.params, b.offsets)]
print("bucket:", len(order), "tensors,", b.flat.numel(), "floats")
# slices that must stay zero: parameters of heads a network never evaluates (their .grad is not written by NerfFn)
must_zero = [(n, off, num) for n, off, num in order if (".sigma." in n and not n.startswith("coarse"))]
# gaps between slices (alignment padding) must stay zero too
gaps = []
for (n0, o0, c0), (n1, o1, c1) in zip(order[:-1], order[1:]):
    if o0 + c0 < o1:
        gaps.append((f"gap after {n0}", o0 + c0, o1 - o0 - c0))
print("watched:", [m[0] for m in must_zero], len(gaps), "gaps")
bad = 0
for it in range(iters):
    h.zero_grad()
    for k, v in h.rays.items():
        if torch.is_tensor(v) and v.requires_grad:
            v.grad = None
    h.fwd_bwd()
    torch.cuda.synchronize()
    for n, off, num in must_zero + gaps:
        sl = b.flat[off:off + num]
        nz = torch.nonzero(sl).flatten()
        if nz.numel():
            bad += 1
            idx = nz[:8].tolist()
            prev = [o for o in order if o[1] + o[2] <= off][-1]
            nxt = [o for o in order if o[1] >= off + num][:1]
            print(f"iter {it}: {n} (bucket offset {off}, {num} floats; before it: {prev[0]} at {prev[1]}+{prev[2]}, after it: {nxt[0][0] if nxt else None}): "
                  f"{nz.numel()} nonzero, first indices {idx}, values {[f'{sl[i].item():.3e}' for i in idx]}", flush=True)
print(f"{prec}: {iters} iterations, {bad} (iteration, slice) pairs with stray writes; MODA_HEAD_STREAMS={os.environ.get('MODA_HEAD_STREAMS', '1')}")
