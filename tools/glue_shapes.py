"""Which PyTorch-native launches does one training step (bf16 mode, eager) still make?  torch.profiler with shapes (this build
records no Python stacks): aten op, input shapes, count per step, device time.  usage: python tools/glue_shapes.py"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from moda_amd.bench_support import TrainHarness

h = TrainHarness(N=2048, S=128, precision="bf16", lr=5e-4)
for _ in range(5):
    h.eager_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        h.eager_step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and ev.self_device_time_total <= 0:
        continue
    if ev.self_device_time_total <= 0:
        continue
    key = (ev.name, str(ev.input_shapes)[:90])
    agg[key][0] += 1
    agg[key][1] += ev.self_device_time_total
tot = 0.0
for (name, shapes), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:28s} {n / 3:6.1f}/step {t / 3:8.1f} us/step  {shapes}")
    tot += t / 3
print(f"total device time of aten ops: {tot:.1f} us per step")
