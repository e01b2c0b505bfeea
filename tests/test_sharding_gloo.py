"""CPU, world_size 2 over gloo: the multi-GPU contract of the path, exercised through the SAME functions bench.py
calls (moda_amd/sharding.py) -- rays shard by rank with no data-path collective; the only exchange is the all-reduce
of the [sum of squared error, ray count] vector, whose result equals the single-process loss over the union of the
rays; the step time is the max over ranks; a training step additionally averages the gradients in one flat bucket.
The images come from the numpy oracle (the HIP path needs a GPU): what is under test is the sharding / reduction code.
Also: bench.py's own launcher (`python bench.py --gpus N` with no WORLD_SIZE) refuses loudly when the GPUs are absent."""
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from moda_amd import sharding, synth
from oracle import moda_oracle as orc
from helpers import oracle_scene

N, S, B = 32, 8, 25
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _images():
    """img_coarse of the whole batch and its target, computed once (numpy oracle) and shared by every rank."""
    scene = oracle_scene(0, B)
    rays = synth.make_rays(0, N, B, rays_per_frame=8)
    img = orc.render_rays(scene, rays, N_samples=S)["img_coarse"].astype(np.float32)
    return rays, img, synth.uniform(2000, "target", (N, 3))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rays, img, target = _images()
    # the slice this rank renders, cut by the function bench.py / a strong-scaling caller uses
    t_rays = {k: torch.from_numpy(v) for k, v in rays.items()}
    t_rays["rays_per_frame"] = 8                                  # non-tensor entries pass through
    shard = sharding.shard_rays(t_rays, rank, world)
    lo, hi = sharding.shard_bounds(N, rank, world)
    assert shard["rays_d"].shape[0] == hi - lo and shard["rays_per_frame"] == 8
    assert torch.equal(shard["bone_rts"], t_rays["bone_rts"][lo:hi])
    # frame-grouped layout (rendering.FRAME_KEYS hold one row per frame): the cut falls on frame boundaries and the per-frame
    # tensors are cut at the same frames -- ray i of the shard still belongs to row i // k
    from moda_amd.rendering import FRAME_KEYS
    f_rays = {k: (v[::8].contiguous() if k in FRAME_KEYS else v) for k, v in t_rays.items()}
    f_shard = sharding.shard_rays(f_rays, rank, world)
    assert lo % 8 == 0 and hi % 8 == 0 and f_shard["rays_d"].shape[0] == hi - lo
    for k in ("bone_rts", "time_embedded", "env_code"):
        assert f_shard[k].shape[0] == (hi - lo) // 8
        assert torch.equal(f_shard[k].repeat_interleave(8, 0), shard[k]), k
    vec = sharding.photometric_sums(torch.from_numpy(img[lo:hi]), torch.from_numpy(target[lo:hi]))
    sharding.allreduce_sums(vec, dist, world)                     # the path's only collective
    tmax = sharding.max_over_ranks(float(rank + 1), "cpu", dist, world)
    seen = sharding.ranks_seen("cpu", dist, world)
    # DDP-style gradient bucket: rank r holds gradient (r + 1) * ones -> mean 1.5 on every rank
    params = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2))]
    params[0].grad = torch.full((3, 5), float(rank + 1))
    params[1].grad = torch.full((7,), 2.0 * (rank + 1))
    n_bucket = sharding.allreduce_gradients(params, dist, world)  # params[2] has no gradient: skipped, as DDP does
    if rank == 0:
        out.put((vec.numpy().tolist(), tmax, seen, n_bucket, params[0].grad[0, 0].item(), params[1].grad[-1].item(),
                 params[2].grad is None))
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
    tot, tmax, seen, n_bucket, g0, g1, g2_none = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    _, img, target = _images()
    single = sharding.photometric_sums(torch.from_numpy(img), torch.from_numpy(target))
    assert sharding.mean_loss(sharding.allreduce_sums(single.clone())) == sharding.mean_loss(single)   # world 1: no-op
    assert tot[1] == N and tmax == 2.0 and seen == 2
    assert abs(tot[0] - float(single[0])) < 1e-6 * abs(float(single[0]))   # rays are independent: sharding changes nothing
    assert n_bucket == 22 and g0 == 1.5 and g1 == 3.0 and g2_none


# ---- world size 8 at the headline batch size (round 4): the shape of the driver's future 8-GPU run -------------------------------
N8, FR8 = 65536 - 3 * 256, 256          # 253 frames of 256 rays: NOT divisible by 8 ranks (uneven shards of whole frames)


def _worker8(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from moda_amd.rendering import FRAME_KEYS
    from moda_amd.autograd import GradBucket
    rays = {k: torch.from_numpy(v) for k, v in synth.make_rays(1000, N8, B, rays_per_frame=FR8).items()}
    img = torch.from_numpy(synth.uniform(3000, "img", (N8, 3)))         # stands in for the render (rays are independent)
    target = torch.from_numpy(synth.uniform(2000, "target", (N8, 3)))
    # per-ray layout: cut anywhere
    pr = sharding.shard_rays(dict(rays, img=img, target=target), rank, world)
    lo, hi = sharding.shard_bounds(N8, rank, world)
    assert pr["rays_d"].shape[0] == hi - lo and torch.equal(pr["bone_rts"], rays["bone_rts"][lo:hi])
    # frame-grouped layout: whole frames per rank, the per-frame tensors cut at the same frames
    fr = {k: (v[::FR8].contiguous() if k in FRAME_KEYS else v) for k, v in rays.items()}
    fr.update(rays_per_frame=FR8, img=img, target=target)
    fs = sharding.shard_rays(fr, rank, world)
    flo, fhi = sharding.shard_bounds(N8, rank, world, align=FR8)
    assert flo % FR8 == 0 and fhi % FR8 == 0 and fs["rays_d"].shape[0] == fhi - flo
    for k in ("bone_rts", "time_embedded", "env_code"):
        assert fs[k].shape[0] == (fhi - flo) // FR8
        assert torch.equal(fs[k].repeat_interleave(FR8, 0), rays[k][flo:fhi]), k
    sums = []
    for sh in (pr, fs):
        v = sharding.photometric_sums(sh["img"], sh["target"])
        sums.append(sharding.allreduce_sums(v, dist, world).tolist())
    counts_pr = sharding.gather_counts(pr["rays_d"].shape[0], "cpu", dist, world)
    counts_fr = sharding.gather_counts(fs["rays_d"].shape[0], "cpu", dist, world)
    seen = sharding.ranks_seen("cpu", dist, world)
    tmax = sharding.max_over_ranks(0.5 + rank, "cpu", dist, world)
    # gradient exchange: GradBucket (the flat buffer the training step's backward kernels add into) + the remaining parameters
    net = [torch.nn.Parameter(torch.zeros(64, 63)), torch.nn.Parameter(torch.zeros(64)), torch.nn.Parameter(torch.zeros(25, 32))]
    rest = [torch.nn.Parameter(torch.zeros(25, 10)), torch.nn.Parameter(torch.zeros(2))]
    bucket = GradBucket(net)
    for i, p_ in enumerate(net):
        p_.grad.add_(float((rank + 1) * (i + 1)))                       # lands in the bucket's views
    rest[0].grad = torch.full((25, 10), float(rank))
    n_b = bucket.all_reduce(dist, world)
    n_r = sharding.allreduce_gradients(rest, dist, world)
    # every rank must now hold the same gradients: min and max over ranks of a checksum agree
    chk = torch.tensor([float(sum(p_.grad.double().sum() for p_ in net + rest[:1]))], dtype=torch.float64)
    cmin, cmax = chk.clone(), chk.clone()
    dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
    dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        out.put(dict(sums=sums, counts_pr=counts_pr, counts_fr=counts_fr, seen=seen, tmax=tmax, n_b=n_b, n_r=n_r,
                     g=[float(p_.grad.flatten()[0]) for p_ in net], g_rest=float(rest[0].grad[0, 0]), rest_none=rest[1].grad is None,
                     cmin=float(cmin), cmax=float(cmax)))
    dist.barrier()
    dist.destroy_process_group()


def test_ray_sharding_world8_headline_batch_uneven_frames():
    """Eight ranks over gloo, 64 768 rays in 253 frames (not a multiple of 8): per-ray and frame-grouped cuts cover every ray once,
    the all-reduced loss vector equals the single-process one in both layouts, per-rank ray counts come back as the shards', the
    gradient bucket and the remaining gradients end up identical (MIN == MAX of a checksum) and equal to the mean over ranks."""
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    r = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    img = torch.from_numpy(synth.uniform(3000, "img", (N8, 3)))
    target = torch.from_numpy(synth.uniform(2000, "target", (N8, 3)))
    single = sharding.photometric_sums(img, target)
    for tot in r["sums"]:
        assert tot[1] == N8 and abs(tot[0] - float(single[0])) < 1e-5 * float(single[0])
    assert r["counts_pr"] == [hi - lo for lo, hi in (sharding.shard_bounds(N8, k, world) for k in range(world))]
    assert r["counts_fr"] == [hi - lo for lo, hi in (sharding.shard_bounds(N8, k, world, align=FR8) for k in range(world))]
    assert sum(r["counts_pr"]) == sum(r["counts_fr"]) == N8 and all(c % FR8 == 0 for c in r["counts_fr"])
    assert max(r["counts_fr"]) - min(r["counts_fr"]) == FR8            # 253 frames over 8 ranks: 31 or 32 frames each
    assert r["seen"] == world and r["tmax"] == 7.5
    assert r["g"] == [4.5, 9.0, 13.5] and r["g_rest"] == 3.5 and r["rest_none"]        # means of (rank + 1) * (i + 1) and of rank
    assert r["n_b"] >= 64 * 63 + 64 + 25 * 32 and r["n_r"] == 250
    assert r["cmin"] == r["cmax"]


def test_shard_bounds_cover_every_ray_once():
    for n in (0, 1, 7, 64, 65536):
        for world in (1, 2, 3, 8):
            cuts = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1
    assert sharding.rank_seed(1000, 3) == 1003


def test_shard_rays_frame_layout_and_errors():
    """Frame-grouped layout: shards are whole frames, also when the frame count does not divide by the world size; a tensor
    that is neither per ray nor (for a FRAME_KEY) per frame is an error, not a silent pass-through."""
    import pytest
    n, k = 40, 8                                                   # 5 frames over 2 ranks: 2 + 3
    rays = {"rays_d": torch.arange(n * 3.).view(n, 3), "bone_rts": torch.arange(5 * 4.).view(5, 4), "rays_per_frame": k,
            "xys": torch.arange(n * 2.).view(n, 2)}
    a, b = sharding.shard_rays(rays, 0, 2), sharding.shard_rays(rays, 1, 2)
    assert a["rays_d"].shape[0] == 16 and b["rays_d"].shape[0] == 24
    assert torch.equal(torch.cat([a["bone_rts"], b["bone_rts"]]), rays["bone_rts"])
    assert torch.equal(torch.cat([a["xys"], b["xys"]]), rays["xys"])
    assert sharding.shard_bounds(40, 1, 2, align=8) == (16, 40)
    with pytest.raises(ValueError):
        sharding.shard_rays(dict(rays, xys=torch.zeros(5, 2)), 0, 2)        # per-frame rows under a per-ray key
    with pytest.raises(ValueError):
        sharding.shard_rays(dict(rays, rays_per_frame=7), 0, 2)
    with pytest.raises(ValueError):
        sharding.shard_rays({"rays_d": torch.zeros(8, 3), "other": torch.zeros(3, 2)}, 0, 2)
    plain = sharding.shard_rays({"rays_d": torch.zeros(9, 3), "near": torch.ones(9, 1), "tag": "x"}, 1, 2)
    assert plain["near"].shape[0] == 5 and plain["tag"] == "x"


def test_bench_self_launch_refuses_without_gpus():
    """`python bench.py --gpus 2` with no launcher starts its own ranks; on a box with fewer GPUs it must say so and exit
    non-zero BEFORE touching a GPU (here: none), instead of asserting on WORLD_SIZE as round 1 did."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    sys.path.insert(0, ROOT)
    import bench
    if bench.visible_gpu_count() >= 2:
        return                                                   # a real multi-GPU box: covered by the driver's scaling run
    assert p.returncode == 2, (p.returncode, p.stderr[-500:])
    assert "requested but only" in p.stderr and p.stdout.strip() == ""
