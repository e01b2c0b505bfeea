"""Ray sharding across the GPUs of one node and the path's only exchange step.

Rays are independent units: rank r renders its own rays with no data-path collective.  What the ranks
exchange is what the reference's DDP trainer exchanges -- the scalar loss statistics, and for a full training step the
gradients (reference nnutils/train_utils.py:101-106 wraps the model in DistributedDataParallel; main.py:20-39 starts
one process per GPU).  The functions here are device-agnostic torch code: bench.py runs them on cuda tensors over
RCCL ("nccl"), tests/test_sharding_gloo.py on CPU tensors over gloo.
"""
import torch

# bench.py sets this (MODA_BENCH_FORCE_NCCL=1) to run every collective of the N > 1 path through an initialised process group of
# ONE rank: the only way a one-GPU box can execute RCCL itself (communicator creation, all_reduce kernels on gfx950, the
# dmabuf-IPC environment) -- tests/test_gpu_multirank.py
COLLECTIVES_AT_WORLD_1 = False


def live(world):
    """Do the collectives run?  With several ranks always; with one rank only on request (see above)."""
    return world > 1 or COLLECTIVES_AT_WORLD_1


def shard_bounds(n_total, rank, world, align=1):
    """Contiguous ray range [lo, hi) of rank `rank` (strong-scaling split of one ray batch).  align = k > 1 cuts only at
    multiples of k (whole frames of the frame-grouped layout); n_total must then be a multiple of k."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    align = int(align)
    if align < 1 or n_total % align:
        raise ValueError(f"{n_total} rays are not whole groups of {align}")
    g = n_total // align
    return (rank * g // world) * align, ((rank + 1) * g // world) * align


def shard_rays(rays, rank, world):
    """The slice of a `rays` dict that rank `rank` renders.

    The dict is the reference's (moda.py:1281-1311): every tensor ray-major on dim 0.  In the frame-grouped layout
    (rays['rays_per_frame'] = k, rendering.FRAME_KEYS may hold ONE row per frame) the cut falls on frame boundaries and the
    per-frame tensors are cut at the same frames, so that ray i of a shard still belongs to row i // k.  A tensor whose
    leading dimension is neither the ray count nor (for FRAME_KEYS) the frame count is an error, not a pass-through: sliced
    rays with an unsliced companion would render the wrong frames silently."""
    from .rendering import FRAME_KEYS
    n = rays['rays_d'].shape[0]
    k = rays.get('rays_per_frame', None)
    k = 1 if k is None else int(k)
    if k < 1 or n % k:
        raise ValueError(f"rays_per_frame={k} does not divide {n} rays")
    lo, hi = shard_bounds(n, rank, world, align=k)
    f = n // k
    out = {}
    for key, v in rays.items():
        if not torch.is_tensor(v) or v.dim() == 0:
            out[key] = v
        elif v.shape[0] == n:
            out[key] = v[lo:hi]
        elif k > 1 and key in FRAME_KEYS and v.shape[0] == f:
            out[key] = v[lo // k:hi // k]
        else:
            raise ValueError(f"rays['{key}']: leading dimension {v.shape[0]} is neither the {n} rays"
                             + (f" nor the {f} frames" if k > 1 and key in FRAME_KEYS else ""))
    return out


def rank_seed(base, rank):
    """Weak scaling: every rank owns its own rays, as every DDP rank of the reference draws its own lines
    (dataloader/frameloader.py:40-45 DistributedSampler)."""
    return int(base) + int(rank)


def photometric_sums(img, target, out=None):
    """[sum of squared colour error, ray count] of this rank's rays, as a 2-vector on img's device."""
    if out is None:
        out = torch.zeros(2, device=img.device, dtype=torch.float32)
    out[0] = (img - target).pow(2).sum()
    out[1] = float(img.shape[0])
    return out


def allreduce_sums(vec, dist=None, world=1):
    """Sum the per-rank statistics vector over all ranks, in place (the path's only collective)."""
    if live(world):
        dist.all_reduce(vec)
    return vec


def mean_loss(vec):
    return float(vec[0] / vec[1])


def allreduce_gradients(params, dist=None, world=1):
    """DDP semantics: every gradient becomes the mean over ranks, exchanged as ONE flat bucket (about 11 MB for MoDA's
    networks; xGMI rings are per-link bound, so one large message beats one per tensor)."""
    if not live(world):
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat /= world
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
    return flat.numel()


def max_over_ranks(seconds, device, dist=None, world=1):
    """The step time every rank reports is the slowest rank's."""
    t = torch.tensor([float(seconds)], device=device, dtype=torch.float64)
    if live(world):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def ranks_seen(device, dist=None, world=1):
    """Number of ranks that actually take part in the collectives (1 added by each)."""
    t = torch.ones(1, device=device, dtype=torch.float32)
    if live(world):
        dist.all_reduce(t)
    return int(round(float(t.item())))


def gather_counts(n_local, device, dist=None, world=1):
    """[rays of rank 0, rays of rank 1, ...] -- what each rank actually rendered per step (an all-reduced one-hot vector: no
    all_gather of python objects, works over RCCL and gloo alike)."""
    t = torch.zeros(max(int(world), 1), device=device, dtype=torch.float64)
    r = dist.get_rank() if (live(world) and dist is not None) else 0
    t[r] = float(n_local)
    if live(world):
        dist.all_reduce(t)
    return [int(round(v)) for v in t.tolist()]
