"""Which torch-native launches does one cfg4 training step contain, and on what shapes?  torch.profiler over one eagerly launched
step of the bench harness: aten operators that launch a kernel, grouped by (op, input shapes), with the autograd node they ran
under when there is one.   usage: python tools/glue_probe.py"""
import os
import sys
import collections

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity

from moda_amd.bench_support import TrainHarness

h = TrainHarness(N=2048, S=128, B=25, precision="bf16")
for _ in range(3):
    h.eager_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    h.eager_step()
    torch.cuda.synchronize()
ev = prof.events()
# kernels launched per CPU op: walk events; keep aten ops that have device time of their own
rows = collections.Counter()
stacks = {}
for e in ev:
    if not e.name.startswith("aten::"):
        continue
    if not e.kernels:
        continue
    kn = ",".join(sorted({k.name.split("<")[0].split("(")[0][-40:] for k in e.kernels}))
    shapes = str(e.input_shapes)[:90]
    key = (e.name, shapes, kn)
    rows[key] += 1
    if key not in stacks and e.stack:
        st = [s for s in e.stack if "moda_amd" in s or "bench_support" in s or "autograd" in s][:3]
        stacks[key] = " <- ".join(x.split("/")[-1][:60] for x in st)
tot = 0
for (name, shapes, kn), c in sorted(rows.items(), key=lambda kv: -kv[1]):
    tot += c
    print(f"{c:3d} {name:22s} {shapes:90s} {kn[:40]:40s} {stacks.get((name, shapes, kn), '')}")
print("torch-native launches in the step:", tot)
