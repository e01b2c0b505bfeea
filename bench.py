"""Benchmark of the MoDA rendering hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one `render_rays` call over this rank's rays (BASELINE.json configs[1]: 65536 rays x 256 samples,
25-bone DQS, 8x256 coarse MLP + 5x64 skin MLP evaluated twice, bf16 MFMA) followed by the photometric loss
and, for N > 1, its RCCL all-reduce.  Rays are sharded by rank with no data-path collective ("weak" scaling:
every GPU renders its own 65536 rays, as every DDP rank of the reference renders its own lines).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC): keep the setting the
# launcher exports even when bench.py is started from a bare environment; read by the HSA runtime at its first use
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic FLOP = 2 x MACs of every nn.Linear the reference evaluates per sample
COARSE_MACS = 601_600
SKIN_MACS = 47_840
FLOP_PER_SAMPLE = 2 * (COARSE_MACS + 2 * SKIN_MACS)   # 1,394,560
PEAK_BF16_TFLOPS = 2500.0                              # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def cpu_baseline(n_rays, S, B):
    """The numpy oracle (a port of the reference's maths, fp32) on a bounded sample of the same workload."""
    from moda_amd import synth
    from oracle import moda_oracle as orc
    from helpers import oracle_scene
    scene = oracle_scene(0, B)
    rays = synth.make_rays(0, n_rays, B, rays_per_frame=256)
    orc.render_rays(scene, {k: v[:16] for k, v in rays.items()}, N_samples=S)   # warm the BLAS threads
    t0 = time.time()
    orc.render_rays(scene, rays, N_samples=S)
    dt = time.time() - t0
    return {"value": n_rays / dt, "unit": "rays/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n_rays} rays x {S} samples of the same workload, numpy fp32 oracle, {dt:.1f} s"}


def train_mode(args, world, rank, local, dist):
    """One training step per rank at the reference recipe's size (scripts/template.sh:7-8,25,28 -> 2048 rays x 128
    samples per GPU): forward + backward through the HIP autograd Functions, DDP-style gradient all-reduce (mean) of
    every trainable tensor in one flat bucket, loss-vector all-reduce, AdamW.  Each rank renders its own rays."""
    import moda_amd
    from moda_amd import synth
    from gpu_helpers import make_models, make_opts, rays_to_gpu
    import gpu_helpers
    gpu_helpers.DEV = f"cuda:{local}"
    N = 2048 if args.rays == 65536 else args.rays
    S = 128 if args.samples == 256 else args.samples
    B = args.bones
    # MoDA's default training configuration (moda.py:60-173): coarse + skin + CSE feature + visibility nets, paired-frame
    # correspondence (dist_corresp), Sinkhorn feature matching (use_ot), all per-ray loss keys present
    models, emb = make_models(0, B, with_feat=True, with_vis=True)
    for m in models.values():
        if isinstance(m, torch.nn.Module):
            m.train()
    models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
    models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
    rays = rays_to_gpu(synth.make_rays(1000 + rank, N, B, rays_per_frame=4))
    rays.update(rays_to_gpu(synth.make_corresp_rays(1000 + rank, N, B, rays_per_frame=4)))
    rays.update(rays_to_gpu(synth.make_feat_rays(1000 + rank, N, rays_per_frame=4)))
    for k in ("bone_rts", "bone_rts_target", "bone_rts_dentrg", "time_embedded", "env_code", "rays_o", "rays_d", "rtk_vec",
              "rtk_vec_target", "rtk_vec_dentrg"):
        rays[k].requires_grad_(True)
    params = [p for m in models.values() if isinstance(m, torch.nn.Module) for p in m.parameters()]
    params += [models["bones_rst"], models["skin_aux"]]
    opts = make_opts(dist_corresp=True, use_corresp=True, use_ot=True)
    bound = np.asarray([0.2, 0.2, 0.2], np.float32)
    loss_buf = torch.zeros(2, device=gpu_helpers.DEV)

    opt = torch.optim.AdamW(params, lr=5e-4, capturable=True)
    vis_neg = torch.empty((1, N * S, 3), device=gpu_helpers.DEV)      # negatives of the visibility loss (loss_utils.py:137)

    def masked_mean(x, m):            # x[m].mean() without the boolean gather (no host sync, graph-capturable)
        m = m.to(x.dtype).expand_as(x)
        return (x * m).sum() / m.sum()

    def fwd_bwd():
        vis_neg.uniform_()
        r = moda_amd.render_rays(models, emb, rays, N_samples=S, perturb=1.0, noise_std=0.0, opts=opts, img_size=512,
                                 obj_bound=bound, rng={"vis_neg_rand": vis_neg})
        sil_m = r["sil_at_samp"] > 0
        # total loss assembled as moda.py:540-640 does (default weights)
        loss = masked_mean(r["img_loss_samp"], sil_m) + 0.1 * masked_mean(r["sil_loss_samp"], r["vis_at_samp"] > 0) \
            + 0.01 * masked_mean(r["frnd_loss_samp"][..., None], sil_m) + 2 * masked_mean(r["flo_loss_samp"], r["sil_at_samp_flo"]) \
            + 0.01 * masked_mean(r["feat_err"], sil_m) + 0.02 * masked_mean(r["proj_err"], sil_m) \
            + r["vis_loss"] + 0.05 * r["frame_cyc_dis"].mean()
        loss.backward()
        return loss.detach()

    def eager_step():
        opt.zero_grad(set_to_none=True)
        loss = fwd_bwd()
        if world > 1:
            grads = [p.grad for p in params if p.grad is not None]
            flat = torch.cat([g.reshape(-1) for g in grads])          # one ~11 MB bucket
            dist.all_reduce(flat)
            flat /= world
            off = 0
            for g in grads:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        loss_buf[0] = loss * N
        loss_buf[1] = float(N)
        if world > 1:
            dist.all_reduce(loss_buf)
        opt.step()
        return loss_buf

    # One rank: the whole step (forward, backward, AdamW: ~1000 launches) is captured once into a HIP graph and replayed --
    # the step is launch-latency-bound when issued eagerly.  Several ranks keep the eager step (collectives in between).
    step = eager_step
    graphed = False
    if world == 1 and not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    eager_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            loss_buf[1] = float(N)              # host scalar: set outside the capture
            with torch.cuda.graph(graph):
                g_loss = fwd_bwd()
                loss_buf[0] = g_loss * N
                opt.step()

            def step():
                graph.replay()
                return loss_buf
            graphed = True
        except Exception as e:      # capture is an optimisation: fall back to the eager step, loudly
            import traceback
            tb = "".join(traceback.format_exc().splitlines(True)[-14:]) if os.environ.get("MODA_BENCH_DEBUG") else ""
            print(f"[bench] HIP graph capture failed ({type(e).__name__}: {str(e).splitlines()[0]}); timing the eager step\n{tb}",
                  file=sys.stderr)
            torch.cuda.synchronize()
            step = eager_step

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lb = step()
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=gpu_helpers.DEV)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        print(json.dumps({
            "metric": "training rays/s (2048 rays x 128 samples per GPU, fwd+bwd+AdamW, fp32)",
            "value": N * world * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg4 training step: {N} rays x {S} samples per GPU, {B} bones, jittered depths, "
                                   "MoDA's default heads (img/sil/flo/feat-match(Sinkhorn)/reproj/vis/feat-render/cycle), "
                                   "gradient and loss all-reduce",
                       "rays_per_gpu": N, "samples_per_ray": S, "bones": B, "sharding": f"rays x{world}",
                       "layout": args.layout},
            "loss": float(lb[0] / lb[1]), "hip_graph": graphed,
            "algorithmic_tflops": 3 * FLOP_PER_SAMPLE * N * S * world * args.steps / dt / 1e12}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--bones", type=int, default=25)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=4096)
    ap.add_argument("--no-graph", action="store_true", help="train mode: time the eagerly launched step instead of the HIP graph")
    ap.add_argument("--layout", default="rays", choices=["rays", "frames"],
                    help="render mode: 'rays' = the reference's layout (per-frame tensors repeated per ray, moda.py:1281-1311); "
                         "'frames' = one bone_rts / code row per frame of 256 rays (rays['rays_per_frame'])")
    ap.add_argument("--mode", default="render", choices=["render", "train"],
                    help="render: the headline metric (forward render_rays, BASELINE configs[1]); "
                         "train: one full training step per rank (configs[3] shape: 2048 rays x 128 samples, fp32 "
                         "forward + backward + AdamW, gradients and loss all-reduced over RCCL)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import moda_amd
    from moda_amd import synth, _lib
    from gpu_helpers import make_models, make_opts, rays_to_gpu
    import gpu_helpers
    gpu_helpers.DEV = f"cuda:{local}"

    if args.mode == "train":
        return train_mode(args, world, rank, local, dist)
    N, S, B = args.rays, args.samples, args.bones
    moda_amd.set_precision(args.precision)
    models, emb = make_models(0, B)
    rays = rays_to_gpu(synth.make_rays(1000 + rank, N, B, rays_per_frame=256))   # each rank owns its own rays
    if args.layout == "frames":
        from moda_amd.rendering import FRAME_KEYS
        assert N % 256 == 0
        rays = {k: (v[::256].contiguous() if k in FRAME_KEYS else v) for k, v in rays.items()}
        rays["rays_per_frame"] = 256
    target = torch.from_numpy(synth.uniform(2000 + rank, "target", (N, 3))).to(gpu_helpers.DEV)
    opts = make_opts()
    loss_buf = torch.zeros(2, device=gpu_helpers.DEV)

    def step():
        res = moda_amd.render_rays(models, emb, rays, N_samples=S, perturb=0, noise_std=0.0, opts=opts, img_size=512)
        loss_buf[0] = (res["img_coarse"] - target).pow(2).sum()
        loss_buf[1] = float(N)
        if world > 1:
            dist.all_reduce(loss_buf)          # RCCL over xGMI: the loss vector, the path's only collective
        return loss_buf

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        fence()
        _lib.PROFILE = {}
        t0 = time.perf_counter()
        for _ in range(args.steps):
            lb = step()
        fence()
        dt = time.perf_counter() - t0
        prof, _lib.PROFILE = _lib.PROFILE, None
    loss = float(lb[0] / lb[1])
    tmax = torch.tensor([dt], device=gpu_helpers.DEV)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # dominant kernel: the fused 8x256 PE+MLP launch, timed by events on its own stream
    tag = f"mlp_fused_W256_{'bf16' if args.precision == 'bf16' else 'f32'}"
    ev = prof.get(tag, [])
    kern_ms = float(np.mean([s.elapsed_time(e) for s, e, _ in ev])) if ev else float("nan")
    units = ev[0][2] if ev else 0
    achieved = 2 * COARSE_MACS * units / (kern_ms * 1e-3) / 1e12 if ev else float("nan")
    peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
    skin_tag = tag.replace("W256", "W64")
    sk = prof.get(skin_tag, [])
    skin_ms = float(np.mean([s.elapsed_time(e) for s, e, _ in sk])) if sk else float("nan")

    # HBM bytes per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
    # (separate passes; FETCH_SIZE doubled per the gfx950 correction), summarised under profiles/ by tools/pmc_summary.py
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        t = json.load(open(tpath)).get(tag)
        if t and t.get("rays") == N and t.get("samples") == S:
            traffic = t["hbm_bytes_per_launch"]

    # secondary figure: the exact-fp32 parity mode (the mode the 1e-4 parity tests run in), smaller batch
    fp32_rays_per_s = None
    if rank == 0 and args.precision == "bf16":
        moda_amd.set_precision("fp32")
        sub = {k: (v[:8192 // (256 if args.layout == "frames" and v.shape[0] != N else 1)] if torch.is_tensor(v) else v)
               for k, v in rays.items()}
        with torch.no_grad():
            moda_amd.render_rays(models, emb, sub, N_samples=S, perturb=0, noise_std=0.0, opts=opts, img_size=512)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            moda_amd.render_rays(models, emb, sub, N_samples=S, perturb=0, noise_std=0.0, opts=opts, img_size=512)
            torch.cuda.synchronize()
            fp32_rays_per_s = 8192 / (time.perf_counter() - t1)
        moda_amd.set_precision(args.precision)

    if rank == 0:
        out = {
            "metric": "rays/s (256 samples/ray, 8x256 MLP, 25 bones)",
            "value": N * world * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"cfg2 cat-pikachiu shapes: {N} rays x {S} samples per GPU, {B}-bone DQS, "
                                   "8x256 coarse + 5x64 skin (x2) MLPs, cycle branch on, forward render_rays + "
                                   "photometric loss all-reduce",
                       "rays_per_gpu": N, "samples_per_ray": S, "bones": B, "sharding": f"rays x{world}",
                       "layout": args.layout},
            "loss": loss,
            "fp32_parity_mode_rays_per_s": fp32_rays_per_s,
            "path_roofline_frac": (N * world * args.steps / dt) * S * FLOP_PER_SAMPLE / 1e12 / (peak * world),
            "roofline": {"bound": "mfma", "kernel": tag, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "ms_per_launch": kern_ms,
                         "flop_per_launch": 2 * COARSE_MACS * units, "skin_mlp_ms_per_launch": skin_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_rays, S, B)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
