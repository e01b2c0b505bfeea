"""Benchmark of the MoDA rendering hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1 without a launcher: starts its own N ranks)
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

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic FLOP = 2 x MACs of every nn.Linear the reference evaluates per sample
COARSE_MACS = 601_600
COARSE_MACS_EXECUTED = 601_600 - 65_536   # xyz_encoding_final (256 x 256, no activation) is folded into dir_encoding on the host
SKIN_MACS = 47_840
FLOP_PER_SAMPLE = 2 * (COARSE_MACS + 2 * SKIN_MACS)   # 1,394,560
PEAK_BF16_TFLOPS = 2500.0                              # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PARITY_GRADE = ("fp16", "bf16x3", "fp32")              # render modes held to the north star's 1e-4 bar against the fp32 oracle


def physical_cores():
    """Physical cores of the host (lscpu: sockets x cores per socket); logical count / 2 if lscpu is unreadable."""
    import subprocess
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {}
        for ln in txt.splitlines():
            if ":" in ln:
                k, v = ln.split(":", 1)
                kv[k.strip()] = v.strip()
        n = int(kv["Socket(s)"]) * int(kv["Core(s) per socket"])
        if n > 0:
            return n, kv.get("Model name", "?")
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2), "?"


def _torch_scene(seed, B):
    """oracle/torch_ref.py model dict (state-dict names -> CPU tensors) of the synthetic scene."""
    from moda_amd import synth
    mp = synth.make_models(seed, B=B)
    conv = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    m = {"coarse": {k: conv(v) for k, v in mp["coarse"].items()}, "bones_rst": conv(mp["bones_rst"]),
         "skin_aux": conv(mp["skin_aux"]), "nerf_skin": {k: conv(v) for k, v in mp["nerf_skin"].items()},
         "rest_pose_code": conv(mp["rest_pose_code"])}
    return m


def cpu_baseline(B, gpu_check=None):
    """the PyTorch-CPU restatement of the path (oracle/torch_ref.py, fp32, op for op the
    reference's maths; pinned to the reference's outputs in tests/test_torch_ref.py) on BASELINE config 1 exactly
    (seed 0, 4096 rays x 64 samples, 25 bones, perturb 0, noise 0), 1 warm-up + median of 3-5 calls.  Reported: the
    figure at torch threads = physical cores (the plan's setting), the single-thread figure, and -- because this
    elementwise-heavy path does not scale over many threads -- a scan over thread counts on a 1024-ray sample whose best
    setting is then timed on config 1 exactly (`cfg1_rays_per_s`: the fastest of these) and on the GPU metric's own workload, 256
    samples per ray, bounded to 1024 rays: that is `value`.  kind 'port'."""
    from moda_amd import synth
    from oracle import torch_ref as tr
    cores, model = physical_cores()
    scene = _torch_scene(0, B)
    prev = torch.get_num_threads()

    def timed(n_rays, S, threads, reps):
        rays = {k: torch.from_numpy(v) for k, v in synth.make_rays(0, n_rays, B, rays_per_frame=256).items()}
        torch.set_num_threads(threads)
        with torch.no_grad():
            t0 = time.perf_counter()
            res = tr.render_rays(scene, rays, S)                       # warm-up
            warm = time.perf_counter() - t0
            if warm > 4.0:
                reps = min(reps, 3)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                res = tr.render_rays(scene, rays, S)
                ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), len(ts), res

    try:
        t_phys, reps_phys, res1 = timed(4096, 64, cores, 5)
        scan = {}
        for th in sorted({1, 4, 16, 32, 64, cores}):
            if th <= cores:
                scan[th] = 1024 / timed(1024, 64, th, 1)[0]
        best_th = max(scan, key=scan.get)
        t_best, reps_best = (t_phys, reps_phys) if best_th == cores else timed(4096, 64, best_th, 5)[:2]
        t_one = 4096 / scan[1] if best_th != 1 else t_best
        t_256, reps256, _ = timed(1024, 256, best_th, 3)
    finally:
        torch.set_num_threads(prev)
    use_best = t_best <= t_phys
    # `value` is the figure on the GPU metric's OWN workload (256 samples per ray: a bounded sample, 1024 of its 65 536 rays) --
    # the one to set beside `value` of this line; BASELINE config 1 (4096 x 64, the reference's CPU-runnable case) travels beside
    # it under its own name (the two were easy to conflate when config 1 was `value`)
    out = {"value": 1024 / t_256, "unit": "rays/s", "cores": cores, "threads": best_th,
           "kind": "port",
           "sample": f"the GPU metric's workload, bounded: 1024 rays x 256 samples, {B} bones, seed 0; oracle/torch_ref.py (PyTorch "
                     f"{torch.__version__} CPU fp32), {best_th} threads (the fastest of a scan), 1 warm-up + median of {reps256}, "
                     f"{t_256:.2f} s per call",
           "cfg1_rays_per_s": 4096 / min(t_best, t_phys),
           "cfg1_sample": f"BASELINE config 1 exactly: 4096 rays x 64 samples, {B} bones, seed 0, "
                          f"{best_th if use_best else cores} threads, 1 warm-up + median of {reps_best if use_best else reps_phys}, "
                          f"{min(t_best, t_phys):.2f} s per call",
           "cpu_model": model, "logical_cpus": os.cpu_count(), "torch": torch.__version__,
           "rays_per_s_at_physical_cores": 4096 / t_phys, "physical_cores_sample": f"config 1, {cores} threads, median of {reps_phys}, {t_phys:.2f} s per call",
           "single_thread_rays_per_s": scan[1] if best_th != 1 else 4096 / t_best,
           "thread_scan_rays_per_s": {str(k): v for k, v in scan.items()}, "thread_scan_sample": "1024 rays x 64 samples, one call after a warm-up",
           "rays_per_s_at_256_samples": 1024 / t_256}
    # how the port's speed relates to the REFERENCE's own render_rays on identical threads: measured where the reference can run
    # (the build container) by tools/cpu_port_vs_reference.py, committed as oracle/port_vs_reference.json
    pv = os.path.join(ROOT, "oracle", "port_vs_reference.json")
    if os.path.exists(pv):
        d = json.load(open(pv))
        out["port_vs_reference_speed"] = d["port_vs_reference_speed"]
        out["port_vs_reference_sample"] = (f"{d['rays']} rays x {d['samples']} samples in the build container ({d['host_cpus']} vCPUs): "
                                           + ", ".join(f"{t} thread(s) {v['port_vs_reference_speed']:.2f}x" for t, v in d["threads"].items())
                                           + "; >= 1: the port is not slower than the reference")
    if gpu_check is not None:      # 'loss match': the HIP path (exact-fp32 mode) on the same config-1 rays vs this CPU result
        w = gpu_check({k: res1[k].numpy() for k in ("img_coarse", "depth_rnd", "sil_coarse")})
        out["gpu_vs_cpu_cfg1_max_rel_err"] = w["fp32"]
        out["gpu_bf16x3_vs_cpu_cfg1_max_rel_err"] = w["bf16x3"]
        out["gpu_fp16_vs_cpu_cfg1_max_rel_err"] = w["fp16"]
    return out


def train_mode(args, world, rank, local, dist):
    """One training step per rank at the reference recipe's size (scripts/template.sh:7-8,25,28 -> 2048 rays x 128
    samples per GPU): forward + backward through the HIP autograd Functions, DDP-style gradient all-reduce (mean) of
    every trainable tensor in one flat bucket, loss-vector all-reduce, AdamW (moda_amd/bench_support.py TrainHarness, the object
    tests/test_gpu_train.py checks).  Each rank renders its own rays.  The number of optimiser steps before the printed loss
    is fixed (--settle-steps + capture warm-up + W + K), so the loss is a deterministic function of the build."""
    from moda_amd import sharding
    from moda_amd.bench_support import TrainHarness, TRAIN_TERMS
    dev = f"cuda:{local}"
    N = 2048 if args.rays is None else args.rays
    S = 128 if args.samples is None else args.samples
    B = args.bones
    strong = args.scaling == "strong"       # one batch of N rays cut across the ranks (default for training: every rank its own N rays)
    h = TrainHarness(N=N, S=S, B=B, precision=args.precision, rank=rank, world=world, dist=dist, lr=args.lr, device=dev,
                     use_fine=args.fine, with_unc=args.unc, strong=strong)
    seen = sharding.ranks_seen(dev, dist, world)
    if seen != world:                        # fail fast: a rank that does not take part in the collectives makes every figure below wrong
        sys.exit(f"[bench] {seen} ranks answer the collectives, WORLD_SIZE is {world}")
    def fence0():
        if sharding.live(world):
            dist.barrier()
        torch.cuda.synchronize()
    # untimed eager steps first (start-of-process stall / clock ramp); the LAST few of them are timed for the record: what the
    # step costs when every launch is issued from Python (`eager_ms_per_step`)
    n_eager = min(5, max(args.settle_steps - 1, 0))
    t0 = None
    first_loss = None
    for i in range(args.settle_steps):
        if n_eager and i == args.settle_steps - n_eager:
            fence0()
            t0 = time.perf_counter()
        h.eager_step()
        if i == 0:
            first_loss = h.loss()            # all-reduced over the ranks: the loss of the INITIAL weights on the union of their rays
    fence0()
    eager_ms = sharding.max_over_ranks(time.perf_counter() - t0, dev, dist, world) / n_eager * 1e3 if n_eager else None
    # The step (forward, backward, AdamW: ~440 launches) is captured once and replayed -- it is launch-latency-bound when issued
    # eagerly.  One rank: one HIP graph.  Several ranks: two graphs around the step's single all-reduce (TrainHarness.capture).
    graphed = False
    if not args.no_graph:
        try:
            h.capture(warm=3)
            graphed = True
        except Exception as e:      # capture is an optimisation: fall back to the eager step, loudly
            import traceback
            tb = "".join(traceback.format_exc().splitlines(True)[-14:]) if os.environ.get("MODA_BENCH_DEBUG") else ""
            print(f"[bench] HIP graph capture failed ({type(e).__name__}: {str(e).splitlines()[0]}); timing the eager step\n{tb}",
                  file=sys.stderr)
            torch.cuda.synchronize()
            h.graph = h.graph_tail = None
            h.graph_form = "eager (capture failed)"

    def fence():
        if sharding.live(world):
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        h.step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h.step()
    fence()
    dt = sharding.max_over_ranks(time.perf_counter() - t0, dev, dist, world)
    # DDP contract: every rank has applied the same averaged gradients, so the ranks' parameters are identical
    per_rank = sharding.gather_counts(h.N, dev, dist, world)
    chk = torch.stack([p.detach().double().sum() for p in h.params]).sum().reshape(1)
    cmin, cmax = chk.clone(), chk.clone()
    if sharding.live(world):
        dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        line = train_line(h, args, world, dt, graphed, seen, n_job=sum(per_rank))
        line["param_checksum_min"], line["param_checksum_max"] = float(cmin), float(cmax)
        line["eager_ms_per_step"], line["graph_form"] = eager_ms, h.graph_form
        line["first_step_loss"] = first_loss
        line["scaling"] = "strong" if strong else "weak"
        line["rays_per_gpu_all_ranks"] = per_rank
        line["collective_backend"] = dist.get_backend() if sharding.live(world) else None
        print(json.dumps(line))
    if sharding.live(world):
        dist.destroy_process_group()


# SURVEY.md 8(d) MACs per sample of the networks a training step evaluates (forward; backward = 2x: dX and dW)
FEAT_MACS, VIS_MACS = 107_392, 30_688


UNC_MACS = 63 * 256 + 6 * 256 * 256 + (256 + 63) * 256 + 256 * 256 + (256 + 32) * 128 + 128 + 256     # nerf_unc 8x256, 63 + 32 in, 1 out


def train_flop_per_step(N, S, grid=8000, fine=False, unc=False):
    """Algorithmic FLOP of one cfg4 training step, SURVEY 8(d)'s convention (2 x MACs of every nn.Linear the reference evaluates;
    backward = 2 x forward): per sample coarse + skin x 2 (backward warp and rest-pose forward skinning, rendering.py:304, 330;
    the target-frame warps of :345-360 re-use the latter) + nerf_feat (rendered features, :174-178) + nerf_vis on the positives
    and on N*S random negatives (loss_utils.py:137-146); per ray the skin net once more (kp_reproj's forward warp of the matched
    point, loss_utils.py:224-270); nerf_feat on the 20^3 matching lattice (:300-312).  The (N x 8000 x 16) cost-volume product
    is not an nn.Linear and is left out, as are PE, skinning, Sinkhorn and compositing."""
    per_sample = COARSE_MACS + 2 * SKIN_MACS + FEAT_MACS + 2 * VIS_MACS
    macs = N * S * per_sample + N * SKIN_MACS + grid * FEAT_MACS + (N * UNC_MACS if unc else 0)
    # hierarchical step: + the no-grad coarse pre-pass on S/2 depths, forward only (coarse + the backward warp's skin net,
    # rendering.py:96-107; SURVEY 8(d) counts the layers the reference evaluates there)
    pre = N * (S // 2) * (COARSE_MACS + SKIN_MACS) if fine else 0
    return 3 * 2 * macs + 2 * pre


def train_line(h, args, world, dt, graphed, seen, n_job=None):
    from moda_amd.bench_support import TRAIN_TERMS
    N, S, B = h.N, h.S, h.B
    flop = train_flop_per_step(N, S, fine=h.use_fine, unc=h.with_unc)
    peak = PEAK_F32_TFLOPS if args.precision == "fp32" else PEAK_BF16_TFLOPS
    cfg = "cfg5" if h.use_fine else "cfg4"
    stage = (f"hierarchical {S // 2}+{S // 2} samples (no-grad coarse pre-pass in the {h.prepass_precision} inference kernels), "
             if h.use_fine else "") + ("uncertainty network + its loss, " if h.with_unc else "")
    ach = flop * world * args.steps / dt / 1e12
    what = {"bf16": "bf16 GEMM operands and saved activations / fp32 accumulate, parameters and gradients",
            "bf16x3": "split-bf16: fp32 storage, hi+lo bf16 operands, 3 MFMAs per product -- ~1e-6 of the fp32 GEMMs",
            "bf16x6": "split-bf16: fp32 storage, hi+mid+lo bf16 operands = the fp32 values, 6 MFMAs per product -- fp32-grade gradients",
            "fp32": "exact fp32"}[args.precision]
    # the roof that binds this step is HBM, not the matrix pipe (its dW / dX GEMMs move their algorithmic bytes at 4-5 TB/s,
    # DESIGN section 9): HBM bytes of one step from the PMC passes of tools/pmc_train_step.sh, if they were taken on THIS build
    hbm = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        from moda_amd.build import source_hash
        key = "train_step_" + args.precision + ("" if (N, S, cfg) == (2048, 128, "cfg4") else f"_{cfg}_{N}x{S}")
        tj = json.load(open(tpath)).get(key)
        if tj:
            fresh = tj.get("kernel_source_sha16") == source_hash()
            gbs = tj["hbm_bytes_per_step"] / (dt / args.steps) / 1e9
            hbm = {"bound": "hbm", "bytes_per_step": tj["hbm_bytes_per_step"] if fresh else None,
                   "achieved": gbs if fresh else None, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0 if fresh else None,
                   "profile_is_of_this_build": fresh, "bytes_per_step_in_profile": tj["hbm_bytes_per_step"],
                   "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over the eagerly launched step, 2 x FETCH + WRITE, all kernels "
                          "(tools/pmc_train_step.sh); divided by this run's ms_per_step"}
    return {
        "roofline_hbm": hbm,
        "metric": f"training rays/s ({N} rays x {S} samples per GPU, fwd+bwd+AdamW, "
                  f"{what})",
        "value": (n_job if n_job is not None else N * world) * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"fp32": "f32"}.get(args.precision, args.precision), "data": "synthetic",
        "config": {"workload": f"{cfg} training step: {N} rays x {S} samples per GPU, {B} bones, jittered depths, {stage}"
                               "MoDA's default heads (img/sil/flo/feat-match(Sinkhorn)/reproj/vis/feat-render/cycle), "
                               "gradient and loss all-reduce",
                   "rays_per_gpu": N, "samples_per_ray": S, "bones": B, "sharding": f"rays x{world}",
                   "layout": args.layout, "lr": args.lr},
        "loss": h.loss(), "loss_terms": dict(zip(TRAIN_TERMS, [float(v) for v in h.terms.tolist()])),
        "optimizer_steps": h.steps_done, "hip_graph": graphed, "n_ranks_seen": seen,
        "roofline": {"bound": "mfma", "kernel": "whole training step (all networks' GEMMs, forward + backward)",
                     "achieved": ach / world, "peak": peak, "unit": "TFLOP/s", "frac": ach / world / peak, "traffic": None,
                     "flop_per_step": flop}}


def visible_gpu_count():
    """GPUs this process tree may use, WITHOUT touching the HIP / HSA runtime (the launcher parent must stay GPU-free: a
    process that has initialised the runtime and then starts ranks holds a context for the whole run, and
    torch.cuda.device_count() may go through hipGetDeviceCount on ROCm builds without amdsmi).  Physical GPUs = KFD topology
    nodes with SIMDs (/sys/class/kfd/kfd/topology/nodes/*/properties: CPUs have simd_count 0), narrowed by the
    *_VISIBLE_DEVICES lists the runtime itself honours."""
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for ln in open(path):
                if ln.startswith("simd_count") and int(ln.split()[1]) > 0:
                    n += 1
        except (OSError, ValueError):
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks ourselves, one
    process per GPU under torch.distributed.run -- the shape of the reference's launch, one command for the node
    (scripts/template-mgpu.sh:22-28, main.py:20-39) -- from a parent that never touches the GPU, pass the ranks' output through
    and exit with the launcher's code (non-zero if any rank failed).  --standalone: the launcher picks its own free
    rendezvous port on 127.0.0.1 (no bind-then-release race)."""
    import subprocess
    have = visible_gpu_count()
    if have < n:
        print(f"[bench] --gpus {n} requested but only {have} visible", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        try:
            json.loads(ln)
            line = ln
        except ValueError:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    if p.returncode != 0:
        return p.returncode
    return 0 if line is not None else 3


def other_configs(args, timed_render):
    """BASELINE.json configs[2..4] on this GPU, reported inside the headline line: cfg3 (36 bones + symmetric-shape branch) and
    cfg5 (hierarchical 128 + 128 samples + CSE feature head) as forward renders of 65536 rays x 256 samples in the bf16 throughput
    mode and in the parity-grade fp16 mode; cfg4 as the full training step at the reference recipe's per-GPU size (2048 rays x 128
    samples) and at the cfg2 batch sharded eight ways (8192 rays x 256 samples: one rank's share); cfg5 as the full
    training step of the reference's last stage (use_fine + feature / uncertainty heads, scripts/template.sh:59) at both sizes."""
    import moda_amd
    from moda_amd import synth
    from moda_amd.bench_support import make_models, make_opts, rays_to_gpu
    out = {}
    N, S = args.rays, args.samples
    for name, B, kw_m, kw_o, fine in (("cfg3_adult7_36bones_symm", 36, dict(perturb_bones=True), dict(symm_shape=True), False),
                                      ("cfg5_ama_fine128+128_cse", 25, dict(with_feat=True), dict(), True)):
        models, emb = make_models(0, B, **kw_m)
        rays = rays_to_gpu(synth.make_rays(1000, N, B, rays_per_frame=256))
        moda_amd.set_precision("bf16")
        t, r = timed_render(models, emb, rays, 10, N_samples=S, perturb=0, noise_std=0.0, opts=make_opts(**kw_o), img_size=512,
                            use_fine=fine)
        # FLOP per ray: `flop_ref` = SURVEY 8(d)'s convention, every nn.Linear the REFERENCE evaluates (hierarchical: the coarse
        # depths twice -- pre-pass and merged final pass); `flop` = what this build evaluates (REUSE_COARSE: every depth once --
        # the pre-pass with its colour branch on S/2 depths, warp + 8x256 on the S/2 importance depths, cycle warp and feature
        # net on all S).  `path_roofline_frac` prices the EXECUTED work; the reference-convention figure travels beside it.
        flop_ref = S * 2 * (COARSE_MACS + 2 * SKIN_MACS + (FEAT_MACS if fine else 0)) + (S // 2 * 2 * (COARSE_MACS + SKIN_MACS) if fine else 0)
        flop = S * 2 * (COARSE_MACS + 2 * SKIN_MACS + FEAT_MACS) if fine else flop_ref
        out[name] = {"rays_per_s": N / t, "ms_per_call": t * 1e3, "rays": N, "samples_per_ray": S, "bones": B, "dtype": "bf16",
                     "img_mean": float(r["img_coarse"].mean()), "path_roofline_frac": N / t * flop / 1e12 / PEAK_BF16_TFLOPS,
                     "path_roofline_frac_reference_flop_convention": N / t * flop_ref / 1e12 / PEAK_BF16_TFLOPS}
        # the same configuration in the parity-grade fp16 mode (cfg5: what its hierarchical pre-pass and feature network run in is
        # rendering.FP16_PREPASS_PRECISION / FP16_FEAT_PRECISION -- DESIGN section 4)
        moda_amd.set_precision("fp16")
        t16, r16 = timed_render(models, emb, rays, 5, N_samples=S, perturb=0, noise_std=0.0, opts=make_opts(**kw_o), img_size=512,
                                use_fine=fine)
        moda_amd.overflow.check()
        out[name]["fp16_mode"] = {"rays_per_s": N / t16, "ms_per_call": t16 * 1e3, "path_roofline_frac": N / t16 * flop / 1e12 / PEAK_BF16_TFLOPS,
                                  "img_max_abs_diff_vs_bf16_mode": float((r16["img_coarse"] - r["img_coarse"]).abs().max())}
        if fine:
            # How this build renders hierarchical rays (rendering.REUSE_COARSE, round 6): the final pass reuses the pre-pass's warp
            # and 8x256 results at the coarse depths and evaluates the importance depths only -- the reference evaluates the coarse
            # depths twice (rendering.py:96-116), and SURVEY 8(d)'s FLOP count, which `path_roofline_frac` keeps, counts both.  In
            # the fp16 mode the feature network, whose rendered output reaches no reference key without rays['feats_at_samp'],
            # runs on the mode's own fp16 kernels (result['feat_rnd'], not a reference key, is then 1.9e-4 from fp32), and the
            # split-bf16 pre-pass warps with the final pass's fp16 skin + warp kernel.  For continuity, round 5's settings:
            from moda_amd import rendering as _R
            out[name]["how"] = ("final pass reuses the pre-pass at the coarse depths (REUSE_COARSE); fp16 mode: pre-pass 8x256 network "
                                "split-bf16, its skin+warp fp16, feature network fp16 (no consumer in this call)")
            saved = (_R.REUSE_COARSE, _R.FP16_FEAT_UNCONSUMED_PRECISION, _R.FP16_X3_PREPASS_WARP)
            try:
                _R.REUSE_COARSE, _R.FP16_FEAT_UNCONSUMED_PRECISION, _R.FP16_X3_PREPASS_WARP = False, "bf16x3", ""
                t5, _ = timed_render(models, emb, rays, 3, N_samples=S, perturb=0, noise_std=0.0, opts=make_opts(**kw_o), img_size=512,
                                     use_fine=fine)
                moda_amd.set_precision("bf16")
                t5b, _ = timed_render(models, emb, rays, 5, N_samples=S, perturb=0, noise_std=0.0, opts=make_opts(**kw_o), img_size=512,
                                      use_fine=fine)
            finally:
                _R.REUSE_COARSE, _R.FP16_FEAT_UNCONSUMED_PRECISION, _R.FP16_X3_PREPASS_WARP = saved
            out[name]["round5_settings"] = {"what": "every merged depth evaluated in the final pass; fp16 mode: feature network and the "
                                                    "pre-pass's skin + warp split-bf16",
                                            "bf16_rays_per_s": N / t5b, "fp16_mode_rays_per_s": N / t5}
        del models, rays, r, r16
    moda_amd.set_precision(args.precision)
    torch.cuda.empty_cache()
    from moda_amd.bench_support import TrainHarness
    train_cfgs = [(f"cfg4_train_step_{p}", 2048, 128, p, False, False, 50, 5, 20) for p in ("bf16", "bf16x6", "fp32")]
    train_cfgs += [(f"cfg4_train_step_{p}_8192x256", 8192, 256, p, False, False, 15, 3, 6) for p in ("bf16", "bf16x6")]
    train_cfgs += [(f"cfg5_train_step_{p}", 2048, 128, p, True, True, 30, 5, 10) for p in ("bf16", "bf16x6")]
    train_cfgs += [(f"cfg5_train_step_{p}_8192x256", 8192, 256, p, True, True, 15, 3, 6) for p in ("bf16", "bf16x6")]
    for name, n_r, n_s, prec, fine, unc, steps, warm, settle in train_cfgs:
        ta = argparse.Namespace(**vars(args))
        ta.precision, ta.steps, ta.warmup, ta.settle_steps = prec, steps, warm, settle
        h = TrainHarness(N=n_r, S=n_s, B=25, precision=prec, lr=ta.lr, use_fine=fine, with_unc=unc)
        for _ in range(ta.settle_steps):
            h.eager_step()
        graphed = True
        try:
            h.capture(warm=3)
        except Exception as e:
            print(f"[bench] {name}: HIP graph capture failed ({type(e).__name__}); timing the eager step", file=sys.stderr)
            h.graph, graphed = None, False
        for _ in range(ta.warmup):
            h.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(ta.steps):
            h.step()
        torch.cuda.synchronize()
        line = train_line(h, ta, 1, time.perf_counter() - t0, graphed, 1)
        out[name] = {k: line[k] for k in ("value", "unit", "ms_per_step", "loss", "loss_terms", "optimizer_steps",
                                          "hip_graph", "roofline", "roofline_hbm", "dtype")}
        out[name]["workload"] = line["config"]["workload"]
        del h
        torch.cuda.empty_cache()
    moda_amd.set_train_precision("fp32")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120, help="timed steps (default: >= 2 s of GPU time, past the DVFS ramp)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=None, help="rays per step (render: 65536; train: 2048 per GPU)")
    ap.add_argument("--samples", type=int, default=None, help="samples per ray (render: 256; train: 128)")
    ap.add_argument("--fine", action="store_true", help="train mode: hierarchical sampling (use_fine: S/2 coarse no-grad + S/2 "
                    "importance samples, rendering.py:91-114) -- the reference's last stage, scripts/template.sh:59")
    ap.add_argument("--unc", action="store_true", help="train mode: the uncertainty network and its loss (--use_unc, moda.py:707-720)")
    ap.add_argument("--bones", type=int, default=25)
    ap.add_argument("--precision", default=None, choices=["bf16", "fp32", "bf16x3", "bf16x6", "fp16"],
                    help="render mode (default fp16): fp16 = the parity-grade throughput mode, the headline -- it meets the north "
                         "star's 1e-4 bar; bf16 = BASELINE configs[1]'s nominal dtype, reported beside it as `throughput_mode` "
                         "(3e-4 off the fp32 mode); bf16x3 / fp32 the slower parity modes.  train mode (default bf16): bf16, "
                         "bf16x3, bf16x6 (fp32-grade gradients), fp32")
    ap.add_argument("--settle", type=float, default=2.0,
                    help="seconds of untimed steps before the W warm-up steps: lets the GPU leave its start-of-process state "
                         "(clock ramp; on this pool the 64-wide kernels run up to 1.8x slower during a process's first second "
                         "of work unless it is the first process on the box).  0 = none.")
    ap.add_argument("--settle-steps", type=int, default=60, help="train mode: untimed eager steps before the graph capture "
                    "(a fixed count, so that the printed loss is a deterministic function of the build)")
    ap.add_argument("--lr", type=float, default=5e-4, help="train mode: AdamW learning rate (reference default, moda.py:90)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the secondary exact-fp32 figure (profiling runs: only the timed workload's kernels)")
    ap.add_argument("--no-graph", action="store_true", help="train mode: time the eagerly launched step instead of the HIP graph")
    ap.add_argument("--layout", default="rays", choices=["rays", "frames"],
                    help="render mode: 'rays' = the reference's layout (per-frame tensors repeated per ray, moda.py:1281-1311); "
                         "'frames' = one bone_rts / code row per frame of 256 rays (rays['rays_per_frame'])")
    ap.add_argument("--scaling", default="both", choices=["both", "weak", "strong"],
                    help="render mode, N > 1: strong = ONE batch of --rays rays cut into contiguous per-rank ranges "
                         "(sharding.shard_rays; the north star's 'shard rays across the 8 GPUs'); weak = every GPU renders its own "
                         "--rays rays (each DDP rank of the reference renders its own lines); both (default) = `value` is the "
                         "strong figure and the line also carries `weak_rays_per_s`")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary figures (cfg3 / cfg4 / cfg5, bf16-vs-fp32 error, "
                    "strong-scaling prediction): profiling runs")
    ap.add_argument("--mode", default="render", choices=["render", "train"],
                    help="render: the headline metric (forward render_rays, BASELINE configs[1]); "
                         "train: one full training step per rank (configs[3] shape: 2048 rays x 128 samples, "
                         "forward + backward + AdamW, gradients and loss all-reduced over RCCL)")
    args = ap.parse_args()

    if args.precision is None:
        args.precision = "bf16" if args.mode == "train" else "fp16"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Test hook (tests/test_gpu_multirank.py): MODA_BENCH_ONE_GPU=1 runs every rank on cuda:0 with the gloo backend, so that the
    # N > 1 code path -- sharding, barriers, the loss / gradient all-reduces, max-over-ranks timing, rank 0's JSON line -- can be
    # executed end to end on a one-GPU box.  RCCL itself needs one device per rank; a real run never sets this.
    one_gpu = os.environ.get("MODA_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    import torch.distributed as dist
    torch.cuda.set_device(local)
    # MODA_BENCH_FORCE_NCCL=1 (tests/test_gpu_multirank.py): initialise the RCCL process group and run every collective of the
    # N > 1 path even with ONE rank -- the only way to execute RCCL itself on a one-GPU box
    force_pg = os.environ.get("MODA_BENCH_FORCE_NCCL") == "1" and "RANK" in os.environ
    if world > 1 or force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import moda_amd
    from moda_amd import synth, sharding, _lib
    from moda_amd import bench_support
    from moda_amd.bench_support import make_models, make_opts, rays_to_gpu
    bench_support.DEV = f"cuda:{local}"
    sharding.COLLECTIVES_AT_WORLD_1 = force_pg and world == 1

    if args.mode == "train":
        return train_mode(args, world, rank, local, dist)
    args.rays = 65536 if args.rays is None else args.rays
    args.samples = 256 if args.samples is None else args.samples
    N, S, B = args.rays, args.samples, args.bones
    if args.precision == "bf16x6":
        sys.exit("bench.py: --precision bf16x6 is a precision of --mode train; the render modes are bf16, fp16, bf16x3 and fp32")
    moda_amd.set_precision(args.precision)
    models, emb = make_models(0, B)
    opts = make_opts()

    def fence():
        if sharding.live(world):
            dist.barrier()
        torch.cuda.synchronize()

    def leg(strong, steps, settle):
        """One timed leg.  strong: ONE batch of N rays, every rank builds it and keeps its contiguous range (sharding.shard_rays:
        the north star's "shard rays across the 8 GPUs"); weak: every rank owns its own N rays (each DDP rank of the reference
        renders its own lines).  W warm-up steps, then exactly `steps` steps between barrier + synchronize brackets, max over ranks."""
        seed = 1000 if strong else sharding.rank_seed(1000, rank)
        rays = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.make_rays(seed, N, B, rays_per_frame=256).items()}
        target = torch.from_numpy(synth.uniform(2000 if strong else sharding.rank_seed(2000, rank), "target", (N, 3)))
        if args.layout == "frames":
            from moda_amd.rendering import FRAME_KEYS
            assert N % 256 == 0
            rays = {k: (v[::256].contiguous() if k in FRAME_KEYS else v) for k, v in rays.items()}
            rays["rays_per_frame"] = 256
        if strong:
            rays["target"] = target                                  # cut with the rays (ray-major like every entry)
            rays = sharding.shard_rays(rays, rank, world)
            target = rays.pop("target")
        rays = {k: (v.to(bench_support.DEV) if torch.is_tensor(v) else v) for k, v in rays.items()}
        target = target.to(bench_support.DEV)
        n_local = rays["rays_d"].shape[0]
        loss_buf = torch.zeros(2, device=bench_support.DEV)

        def step(collective=True):
            res = moda_amd.render_rays(models, emb, rays, N_samples=S, perturb=0, noise_std=0.0, opts=opts, img_size=512)
            sharding.photometric_sums(res["img_coarse"], target, out=loss_buf)
            if not collective:
                return loss_buf
            return sharding.allreduce_sums(loss_buf, dist, world)     # RCCL over xGMI: the loss vector, the path's only collective

        with torch.no_grad():
            # the settle phase runs by the clock, so the ranks do DIFFERENT numbers of steps: no collective in it (a rank one
            # all-reduce ahead of the others would pair it with their next one and leave the process group out of step)
            t_settle = time.perf_counter()
            n_settle = 0
            while settle > 0 and time.perf_counter() - t_settle < settle:
                step(collective=False)
                torch.cuda.synchronize()
                n_settle += 1
            for _ in range(args.warmup):
                step()
            fence()
            _lib.PROFILE = {}
            t0 = time.perf_counter()
            for _ in range(steps):
                lb = step()
            fence()
            dt = time.perf_counter() - t0
            prof, _lib.PROFILE = _lib.PROFILE, None
        per_rank = sharding.gather_counts(n_local, bench_support.DEV, dist, world)
        return {"rays": rays, "target": target, "n_local": n_local, "n_job": N if strong else N * world, "n_settle": n_settle,
                "dt": sharding.max_over_ranks(dt, bench_support.DEV, dist, world), "loss": sharding.mean_loss(lb), "prof": prof,
                "per_rank": per_rank, "steps": steps}

    # N = 1: one leg (strong and weak coincide).  N > 1: `value` is the STRONG figure -- one batch of --rays rays cut across the
    # ranks, the north star's sharding -- and the same line carries the weak figure (every rank its own --rays rays) beside it,
    # timed over a third of the steps.  --scaling weak|strong restricts the run to one leg (`value` is then that leg's).
    strong = args.scaling in ("both", "strong") or world == 1
    main_leg = leg(strong, args.steps, args.settle)
    weak_leg = None
    if world > 1 and args.scaling == "both":
        weak_leg = leg(False, max(args.steps // 3, 3), 0.0)
    rays, target, n_local, n_job, n_settle = (main_leg[k] for k in ("rays", "target", "n_local", "n_job", "n_settle"))
    dt, loss, prof = main_leg["dt"], main_leg["loss"], main_leg["prof"]
    seen = sharding.ranks_seen(bench_support.DEV, dist, world)
    if seen != world:
        sys.exit(f"[bench] {seen} ranks answer the collectives, WORLD_SIZE is {world}")

    # dominant kernel: the fused 8x256 PE+MLP launch, timed by events on its own stream
    tag = "mlp_fused_W256_" + {"bf16": "bf16", "fp16": "f16", "bf16x3": "bf16x3"}.get(args.precision, "f32")

    def tag_ms(t):
        ev_ = prof.get(t, [])
        return (float(np.mean([s.elapsed_time(e) for s, e, _ in ev_])), ev_[0][2]) if ev_ else (float("nan"), 0)
    if os.environ.get("MODA_BENCH_TRACE") and rank == 0:      # diagnostic: per-step kernel times (start-up transients)
        for t_, ev_ in prof.items():
            ts_ = [round(s_.elapsed_time(e_), 2) for s_, e_, _ in ev_]
            print(f"[bench trace] {t_}: {ts_}", file=sys.stderr)
    kern_ms, units = tag_ms(tag)
    achieved = 2 * COARSE_MACS * units / (kern_ms * 1e-3) / 1e12 if units else float("nan")
    peak = PEAK_F32_TFLOPS if args.precision == "fp32" else PEAK_BF16_TFLOPS     # fp16 / bf16 MFMAs share one dense peak
    other_ms = {t: tag_ms(t)[0] for t in sorted(prof) if t != tag}

    # HBM bytes per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
    # (separate passes; FETCH_SIZE doubled per the gfx950 correction), summarised under profiles/ by tools/pmc_summary.py
    # `traffic` is a measurement of THIS build or nothing: the profile carries the hash of the kernel sources it was collected on
    # (moda_amd.build.source_hash, stamped by tools/pmc_summary.py); another build's figure is reported as stale beside a null.
    traffic = None
    traffic_src = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        from moda_amd.build import source_hash
        tj = json.load(open(tpath))
        t = tj.get(tag)
        stamp = (tj.get("_stamp") or {}).get("kernel_source_sha16")
        if t and t.get("rays") == n_local and t.get("samples") == S:
            fresh = stamp is not None and stamp == source_hash()
            traffic = t["hbm_bytes_per_launch"] if fresh else None
            traffic_src = {"file": "profiles/traffic.json", "kernel_source_sha16_of_profile": stamp,
                           "kernel_source_sha16_of_this_tree": source_hash(), "fresh": fresh,
                           "hbm_bytes_per_launch_in_profile": t["hbm_bytes_per_launch"],
                           "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/profile_round.sh), "
                                  "2 x FETCH + WRITE per the gfx950 correction; collected in a separate profiled run, not in this one"}

    # secondary figures (rank 0, untimed by the driver's metric): the exact-fp32 parity mode (the mode the 1e-4 parity tests run
    # in) and what the bf16 mode costs in accuracy against it, on the first 8192 rays of this rank
    fp32_rays_per_s = None
    bf16_err = None
    x3 = x16 = None
    render_kw = dict(N_samples=S, perturb=0, noise_std=0.0, opts=opts, img_size=512)

    def timed_render(mdl, em, rr, reps, **kw):
        with torch.no_grad():
            for _ in range(2):
                res_ = moda_amd.render_rays(mdl, em, rr, **kw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                res_ = moda_amd.render_rays(mdl, em, rr, **kw)
            torch.cuda.synchronize()
        return (time.perf_counter() - t1) / reps, res_

    n_sub = min(8192, n_local)
    per_frame = (lambda v: args.layout == "frames" and v.shape[0] != n_local)
    sub = {k: (v[:n_sub // (256 if per_frame(v) else 1)] if torch.is_tensor(v) else v) for k, v in rays.items()}
    # (N = 1 only: with several ranks these legs would keep rank 0 busy for ~15 s while the others sit in the process group's
    #  teardown, and they describe one GPU anyway)
    xb = None
    if rank == 0 and world == 1 and args.precision in ("bf16", "fp16") and not args.no_fp32 and n_sub > 0:
        keys = ("img_coarse", "depth_rnd", "sil_coarse", "xyz_canonical_vis", "frame_cyc_dis")
        moda_amd.set_precision("fp32")
        t_f, r_f = timed_render(models, emb, sub, 3, **render_kw)

        def parity(mode, reps, what):
            """A mode on the WHOLE timed batch (BASELINE config 2's 65 536 rays), its distance from the exact-fp32 mode on the
            batch's first n_sub rays: max|a - b| / max|b| per output and the per-element figure of tests/helpers.elem_err."""
            moda_amd.set_precision(mode)
            t_m, r_m = timed_render(models, emb, rays, reps, **render_kw)
            err, elem = {}, {}
            for k in keys:
                a, b = r_m[k][:n_sub], r_f[k]
                d, mx = (a - b).abs(), b.abs().max().clamp_min(1e-30)
                err[k] = float(d.max() / mx)
                elem[k] = float((d / (1e-4 * b.abs() + 1e-5 * mx)).max())
            return {"rays_per_s": n_local / t_m, "ms_per_step": t_m * 1e3, "rays": n_local, "mode": what,
                    "max_rel_err_vs_fp32_mode": err, "per_element_figure_vs_fp32_mode": elem,
                    "error_sample": f"first {n_sub} rays of the timed batch",
                    "loss": float((r_m["img_coarse"][:n_sub] - target[:n_sub]).pow(2).sum() / n_sub)}
        # fp16 operands in the hot loop: the parity-grade mode at the throughput mode's speed class (the headline since round 6)
        x16 = parity("fp16", 10, "fp16 (fp16 MFMA operands in the 8x256 network and the fused skin+warp kernels, fp32 accumulate; "
                                 "overflow reported, never saturated)")
        moda_amd.overflow.check()
        # bf16 operands: BASELINE configs[1]'s nominal dtype -- the throughput mode, 1e-4 .. 3e-4 off the fp32 mode
        xb = parity("bf16", 10, "bf16 (bf16 MFMA operands, fp32 accumulate): throughput mode, NOT within the 1e-4 bar")
        # split-bf16 (operands as bf16 hi + lo, three MFMAs per product): ~1e-6 of the fp32 mode
        x3 = parity("bf16x3", 3, "bf16x3 (split-bf16 operands, 3 MFMAs per product, fp32 accumulate)")
        moda_amd.set_precision(args.precision)
        fp32_rays_per_s = n_sub / t_f
        bf16_err = dict(xb["max_rel_err_vs_fp32_mode"])
        bf16_err["loss_bf16"] = xb["loss"]
        bf16_err["loss_fp32"] = float((r_f["img_coarse"] - target[:n_sub]).pow(2).sum() / n_sub)
        bf16_err["sample"] = f"first {n_sub} rays of the timed batch; max|bf16 - fp32| / max|fp32| per output"

    # strong-scaling prediction from ONE GPU: a rank of an 8-GPU strong-scaling job renders 1/8 of the batch; its time against
    # 1/8 of the full batch's time is the efficiency the fixed per-call costs (code folds, table kernels, small launches) allow
    strong_pred = None
    if rank == 0 and world == 1 and not args.no_configs and n_local >= 8 * 256:
        n8 = n_local // 8 // 256 * 256
        sub8 = {k: (v[:n8 // (256 if per_frame(v) else 1)] if torch.is_tensor(v) else v) for k, v in rays.items()}
        t_full = dt / args.steps
        t_8, _ = timed_render(models, emb, sub8, 20, **render_kw)
        strong_pred = {"rays_full": n_local, "ms_full": t_full * 1e3, "rays_eighth": n8, "ms_eighth": t_8 * 1e3,
                       "predicted_efficiency_8gpu": (t_full * n8 / n_local) / t_8,
                       "note": "one-GPU measurement: t(full batch) / 8 over t(an eighth of the batch); excludes the 8-byte loss "
                               "all-reduce"}

    # the other BASELINE configurations, so that the driver's record carries them
    configs = None
    if rank == 0 and world == 1 and not args.no_configs and args.precision in ("bf16", "fp16"):
        configs = other_configs(args, timed_render)

    def gpu_cfg1_check(cpu_res):
        """BASELINE config 1 on the HIP path (exact-fp32 mode, and the split-bf16 parity-grade mode) against the CPU baseline's
        own outputs: max relative error over img / depth / sil."""
        r1 = rays_to_gpu(synth.make_rays(0, 4096, B, rays_per_frame=256))
        worst = {}
        for mode in ("fp32", "bf16x3", "fp16"):
            moda_amd.set_precision(mode)
            with torch.no_grad():
                g = moda_amd.render_rays(models, emb, r1, N_samples=64, perturb=0, noise_std=0.0, opts=opts, img_size=512)
            w = 0.0
            for k, ref in cpu_res.items():
                a = g[k].cpu().numpy().astype(np.float64)
                w = max(w, float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)))
            worst[mode] = w
        moda_amd.set_precision(args.precision)
        return worst

    if rank == 0:
        out = {
            "metric": "rays/s (256 samples/ray, 8x256 MLP, 25 bones)",
            "value": n_job * args.steps / dt,
            "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            # (since round 4 the N > 1 headline is the STRONG figure -- one batch cut across the ranks; rounds 1-3 printed the weak
            #  one, now `weak_rays_per_s`, an indicative shorter leg: steps / 3, no settle phase.  BASELINE.md section 4)
            "value_semantics": "one GPU" if world == 1 else ("strong scaling: one batch of --rays rays cut across the ranks" if strong
                                                              else "weak scaling: --rays rays per rank"),
            "weak_rays_per_s": (None if weak_leg is None else weak_leg["n_job"] * weak_leg["steps"] / weak_leg["dt"]) if world > 1
            else n_job * args.steps / dt,
            "weak_leg": None if weak_leg is None else {"rays_per_gpu": weak_leg["per_rank"], "rays_per_step": weak_leg["n_job"],
                                                        "steps": weak_leg["steps"], "ms_per_step": weak_leg["dt"] / weak_leg["steps"] * 1e3,
                                                        "loss": weak_leg["loss"]},
            "dtype": {"fp32": "f32", "fp16": "f16"}.get(args.precision, args.precision), "data": "synthetic",
            # `value` at matched output (north star: within 1e-4 rel of the fp32 reference path): the timed run itself when its
            # mode is parity-grade (fp16, bf16x3, fp32 -- tests/test_gpu_parity.py pins each against the fp32 oracle at THIS
            # shape), else the fp16 mode's figure on the same batch.  `throughput_mode`: the bf16 figure (BASELINE configs[1]'s
            # nominal dtype) with its distance from the fp32 mode.
            "value_at_parity": (n_job * args.steps / dt) if args.precision in PARITY_GRADE else (None if x16 is None else x16["rays_per_s"]),
            "dtype_at_parity": {"fp32": "f32", "fp16": "f16"}.get(args.precision, args.precision) if args.precision in PARITY_GRADE
            else (None if x16 is None else "f16"),
            "throughput_mode": None if xb is None else {"dtype": "bf16", "rays_per_s": xb["rays_per_s"], "ms_per_step": xb["ms_per_step"],
                                                        "max_rel_err_vs_fp32_mode": xb["max_rel_err_vs_fp32_mode"],
                                                        "within_1e-4": max(xb["max_rel_err_vs_fp32_mode"].values()) < 1e-4},
            "config": {"workload": f"cfg2 cat-pikachiu shapes: {N} rays x {S} samples "
                                   f"{'in all, cut into per-GPU ranges' if strong else 'per GPU'}, {B}-bone DQS, "
                                   "8x256 coarse + 5x64 skin (x2) MLPs, cycle branch on, forward render_rays + "
                                   "photometric loss all-reduce",
                       "rays_per_gpu": n_local, "rays_per_gpu_all_ranks": main_leg["per_rank"], "rays_per_step": n_job,
                       "samples_per_ray": S, "bones": B, "sharding": f"rays x{world} ({'strong' if strong else 'weak'})",
                       "layout": args.layout},
            "loss": loss, "n_ranks_seen": seen, "settle_s": args.settle, "settle_steps": n_settle,
            "fp32_parity_mode_rays_per_s": fp32_rays_per_s,
            "parity_mode_rays_per_s": None if x16 is None else x16["rays_per_s"],
            "parity_mode": x16,
            "parity_mode_bf16x3": x3,
            "bf16_vs_fp32_max_rel_err": bf16_err,
            "strong_scaling_prediction": strong_pred,
            "configs": configs,
            "path_roofline_frac": (n_job * args.steps / dt) * S * FLOP_PER_SAMPLE / 1e12 / (peak * world),
            "roofline": {"bound": "mfma", "kernel": tag, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src, "ms_per_launch": kern_ms,
                         "flop_per_launch": 2 * COARSE_MACS * units, "other_kernels_ms_per_launch": other_ms,
                         # the kernel executes fewer MACs than the reference's layer list has (fold of the activation-free
                         # xyz_encoding_final into dir_encoding): matrix-pipe throughput actually sustained, for the record
                         "executed_tflops": 2 * COARSE_MACS_EXECUTED * units / (kern_ms * 1e-3) / 1e12 if units else float("nan"),
                         "executed_frac_of_peak": 2 * COARSE_MACS_EXECUTED * units / (kern_ms * 1e-3) / 1e12 / peak if units else float("nan")},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, gpu_check=gpu_cfg1_check)
        out["collective_backend"] = dist.get_backend() if sharding.live(world) else None
        print(json.dumps(out))
    if sharding.live(world):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
