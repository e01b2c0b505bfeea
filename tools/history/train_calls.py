"""Lists the network-level library calls of one training step of bench.py's train mode (shapes, flags).
usage: python tools/train_calls.py"""
import collections
import sys

import torch

sys.argv = ["bench.py", "--mode", "train", "--precision", "bf16", "--steps", "1", "--warmup", "1", "--no-graph", "--settle", "0"]
import moda_amd._lib as L

orig = L.call
log = []
on = [False]


def spy(name, *a):
    if on[0] and name.startswith("moda_nerf_train"):
        d = a[0]._obj
        log.append((name, d.W, d.D, d.M, d.R1, d.Rd, d.C1, d.Cd, d.n_out, d.sigma_only, d.raw_feat, d.reserved))
    elif on[0]:
        log.append((name,))
    return orig(name, *a)


L.call = spy
import bench

_orig_sync = torch.cuda.synchronize
count = [0]


def main():
    on[0] = True
    try:
        bench.main()
    finally:
        on[0] = False


main()
c = collections.Counter(log)
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print(v, k)
