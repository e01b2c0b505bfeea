"""The matching head's Sinkhorn chain at cfg4's size (2048 pixels x 8000 lattice points, bf16 matrix): one persistent launch each
way (moda_match_sinkhorn) against 78 per-sweep launches, eager and graph-replayed, interleaved in one process."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import numpy as np, torch
import moda_amd
from moda_amd import synth, autograd as A
from moda_amd.bench_support import T

N, G = 2048, 8000
f = T(synth.normal(44, "ps/f", (N, 16))); v = T(synth.normal(44, "ps/v", (G, 16))); q = T(np.float32(0.2) * synth.normal(44, "ps/q", (G, 3)))
gp = T(synth.normal(44, "ps/g", (N, 3))); kap = T(np.asarray([1 / 0.03], np.float32))
moda_amd.set_train_precision("bf16")


def head():
    fg, vg = f.clone().requires_grad_(True), v.clone().requires_grad_(True)
    pg = A.FeatMatchFn.apply(A.NormalizeFn.apply(fg), A.NormalizeFn.apply(vg), q, kap, True)[0]
    (pg * gp).sum().backward()
    return pg


def timed(persist, graph):
    A.SINKHORN_PERSIST = persist
    for _ in range(3):
        head()
    torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            head()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            head()
        run = g.replay
    else:
        run = head
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        s_ = torch.cuda.Event(enable_timing=True); e_ = torch.cuda.Event(enable_timing=True)
        s_.record(); run(); e_.record(); torch.cuda.synchronize(); ts.append(s_.elapsed_time(e_))
    return float(np.median(ts))


for rnd in range(2):
    for graph in (False, True):
        a, b = timed(False, graph), timed(True, graph)
        print(f"matching head fwd+bwd ({'graph replay' if graph else 'eager'}): per-sweep launches {a * 1e3:.0f} us, persistent {b * 1e3:.0f} us, saved {(a - b) * 1e3:.0f} us")
