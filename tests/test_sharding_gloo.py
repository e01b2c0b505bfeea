"""CPU, world_size 2 over gloo: the multi-GPU contract of the path -- rays shard by rank with no data-path
collective, and the only exchange is the all-reduce of the [sum of squared error, ray count] loss vector, whose
result equals the single-process loss over the union of the rays.  The renderer here is the numpy oracle (the
HIP path needs a GPU); what is under test is the sharding / reduction logic bench.py uses."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from moda_amd import synth
from oracle import moda_oracle as orc
from helpers import oracle_scene

N, S, B = 32, 8, 25


def _loss_parts(rank, world):
    scene = oracle_scene(0, B)
    rays = synth.make_rays(0, N, B, rays_per_frame=8)
    lo, hi = rank * N // world, (rank + 1) * N // world          # contiguous ray ranges
    shard = {k: v[lo:hi] for k, v in rays.items()}
    target = synth.uniform(2000, "target", (N, 3))[lo:hi]
    img = orc.render_rays(scene, shard, N_samples=S)["img_coarse"]
    return np.asarray([((img - target) ** 2).sum(), hi - lo], np.float64)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    v = torch.from_numpy(_loss_parts(rank, world))
    dist.all_reduce(v)                                            # the path's only collective
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                      # bench.py's max-over-ranks timing reduction
    if rank == 0:
        out.put((v.numpy().tolist(), float(t)))
    dist.barrier()
    dist.destroy_process_group()


def test_ray_sharding_loss_allreduce_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    (tot, tmax) = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single = _loss_parts(0, 1)
    assert tot[1] == N and tmax == 2.0
    assert abs(tot[0] - single[0]) < 1e-6 * abs(single[0])      # rays are independent: sharding changes nothing
