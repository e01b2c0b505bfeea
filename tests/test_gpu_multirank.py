"""GPU (-m gpu): bench.py's N > 1 code path executed end to end on ONE GPU -- two ranks under torch.distributed.run, both on
cuda:0, gloo instead of RCCL (MODA_BENCH_ONE_GPU=1; RCCL needs one device per rank, which a one-GPU box cannot give).  What runs
is everything else of the multi-GPU contract: per-rank ray sets (weak) or the contiguous cut of one batch (strong), the barrier
+ synchronize bracket, the loss-vector all-reduce, DDP-style gradient exchange in train mode, the max over ranks, and rank 0's
single JSON line.  Reference launch shape: scripts/template-mgpu.sh:22-28, main.py:20-39."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, n=2, settle="0"):
    env = dict(os.environ, MODA_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--no-cpu-baseline", "--no-fp32",
           "--no-configs", "--settle", settle] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]               # rank 0 alone prints, once
    return json.loads(lines[0])


def test_two_ranks_weak_and_strong_render():
    base = ["--rays", "4096", "--samples", "64", "--steps", "3", "--warmup", "1"]
    one = _run(base, n=1)
    both = _run(base)                                        # the default: value = strong, weak beside it
    weak = _run(base + ["--scaling", "weak"])
    strong = _run(base + ["--scaling", "strong"])
    for d in (both, weak, strong):
        assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2
        assert d["value"] > 0 and d["unit"] == "rays/s" and "roofline" in d
    assert one["scaling"] == "strong" and one["weak_rays_per_s"] == one["value"]        # N = 1: the two coincide
    assert both["scaling"] == "strong" and strong["scaling"] == "strong" and weak["scaling"] == "weak"
    assert weak["config"]["rays_per_step"] == 8192 and weak["config"]["rays_per_gpu_all_ranks"] == [4096, 4096]
    for d in (both, strong):
        assert d["config"]["rays_per_step"] == 4096 and d["config"]["rays_per_gpu_all_ranks"] == [2048, 2048]
        # strong scaling renders the SAME 4096 rays as the single rank: the all-reduced loss is the single-rank loss
        assert abs(d["loss"] - one["loss"]) < 1e-6 * abs(one["loss"]), (d["loss"], one["loss"])
    # the default line carries the weak leg too: every rank its own 4096 rays
    assert both["weak_rays_per_s"] > 0 and both["weak_leg"]["rays_per_gpu"] == [4096, 4096] and both["weak_leg"]["rays_per_step"] == 8192
    assert abs(both["weak_leg"]["loss"] - weak["loss"]) < 1e-6 * abs(weak["loss"])
    assert strong["weak_rays_per_s"] is None and weak["weak_leg"] is None
    # weak scaling: rank 0 renders the single rank's rays, rank 1 its own; the loss is the mean over both sets
    assert weak["loss"] != one["loss"] and 0.1 < weak["loss"] < 0.5


def test_clock_driven_settle_phase_keeps_the_ranks_in_step():
    """The driver runs bench.py with its DEFAULT --settle (seconds of untimed steps, by the clock), so the ranks may do different
    numbers of settle steps: that phase must not contain a collective.  Round 4 found the loss all-reduce in it -- a race: the
    blocking collective kept the ranks within a step of each other, but a rank whose clock ran out one step later stayed one
    all-reduce behind for the rest of the run, and its last one met the others' closing barrier (mismatched sizes; over RCCL a
    hang).  Four free-running ranks on one GPU here; the strong-scaling loss must be the single rank's."""
    base = ["--rays", "4096", "--samples", "64", "--steps", "3", "--warmup", "1"]
    one = _run(base, n=1)
    d = _run(base, n=4, settle="0.5")
    assert d["n_gpus"] == 4 and d["n_ranks_seen"] == 4 and d["config"]["rays_per_gpu_all_ranks"] == [1024] * 4
    assert abs(d["loss"] - one["loss"]) < 1e-6 * abs(one["loss"]), (d["loss"], one["loss"])
    assert abs(d["weak_leg"]["loss"] - _run(base + ["--scaling", "weak"], n=4)["loss"]) < 1e-6 * abs(d["weak_leg"]["loss"])


def test_eight_ranks_on_one_gpu_strong_loss_equals_single_rank():
    """The driver's future `--gpus 8` run, executed on one GPU over gloo: one batch of 8192 rays cut into eight contiguous shards
    (1024 rays each), the union loss equal to the single-rank loss, the weak leg beside it."""
    base = ["--rays", "8192", "--samples", "64", "--steps", "2", "--warmup", "1"]
    one = _run(base, n=1)
    d = _run(base, n=8)
    assert d["n_gpus"] == 8 and d["n_ranks_seen"] == 8 and d["scaling"] == "strong"
    assert d["config"]["rays_per_gpu_all_ranks"] == [1024] * 8 and d["config"]["rays_per_step"] == 8192
    assert abs(d["loss"] - one["loss"]) < 1e-6 * abs(one["loss"]), (d["loss"], one["loss"])
    assert d["weak_leg"]["rays_per_gpu"] == [8192] * 8 and d["weak_leg"]["rays_per_step"] == 65536 and d["weak_rays_per_s"] > 0


def test_two_ranks_training_step_exchanges_gradients():
    """Since round 5 the multi-rank step is graph-captured too: two HIP graphs around the step's ONE all-reduce (gradient bucket
    + loss sums).  settle 2 + capture warm-up 3 + warm-up 1 + 2 timed steps = 8 optimiser steps."""
    d = _run(["--mode", "train", "--rays", "256", "--samples", "32", "--steps", "2", "--warmup", "1", "--settle-steps", "2"])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2
    assert d["hip_graph"] is True and d["graph_form"] == "two graphs around one eager all-reduce", d["graph_form"]
    assert d["eager_ms_per_step"] > 0 and d["collective_backend"] == "gloo"
    assert d["optimizer_steps"] == 8 and d["loss"] == d["loss"] and 0.5 < d["loss"] < 5
    # DDP contract: every rank applies the SAME averaged gradients from the first step on, so the ranks' weights
    # stay identical -- the min and the max over ranks of a parameter checksum agree after the five steps
    assert d["param_checksum_min"] == d["param_checksum_max"], (d["param_checksum_min"], d["param_checksum_max"])


def test_two_ranks_graphed_training_step_equals_the_eager_step():
    """The two-graph form against the eagerly launched multi-rank step from the same state: same loss after the same number of
    optimiser steps (to the atomics' order), identical parameter checksums across ranks in both."""
    args = ["--mode", "train", "--rays", "256", "--samples", "32", "--steps", "2", "--warmup", "1", "--settle-steps", "2"]
    g = _run(args)
    e = _run(args + ["--no-graph"])
    assert e["hip_graph"] is False and e["graph_form"] == "eager"
    assert e["optimizer_steps"] == g["optimizer_steps"] - 3              # the capture's three warm-up steps
    for d in (g, e):
        assert d["param_checksum_min"] == d["param_checksum_max"]
    g2 = _run(args)
    assert abs(g["loss"] - g2["loss"]) < 2e-3 * abs(g["loss"])


def test_eight_ranks_training_step_on_the_union_of_the_single_rank_batch():
    """VERDICT r05 #6a: the 8-rank TRAINING step (the driver's future `--gpus 8 --mode train`), on one GPU over gloo, strong form:
    ONE batch of 2048 rays cut into eight contiguous shards of 256.  (1) every rank applies the same averaged gradients: the
    parameter checksums' min and max over ranks agree after all steps; (2) the all-reduced loss of the INITIAL weights equals the
    single-rank step's over the same 2048 rays -- to the batch statistics each rank takes over its own rays, as every DDP rank of the
    reference does (silhouette class balance, flow-confidence mean, masked means: rendering.py:535-555), and the
    visibility loss's per-rank negatives."""
    args = ["--mode", "train", "--rays", "2048", "--samples", "32", "--steps", "2", "--warmup", "1", "--settle-steps", "2",
            "--scaling", "strong"]
    one = _run(args, n=1)
    d = _run(args, n=8)
    assert d["n_gpus"] == 8 and d["n_ranks_seen"] == 8 and d["scaling"] == "strong"
    assert d["rays_per_gpu_all_ranks"] == [256] * 8 and one["rays_per_gpu_all_ranks"] == [2048]
    assert d["hip_graph"] is True and d["graph_form"] == "two graphs around one eager all-reduce"
    assert d["param_checksum_min"] == d["param_checksum_max"], (d["param_checksum_min"], d["param_checksum_max"])
    assert d["optimizer_steps"] == one["optimizer_steps"] == 8
    assert abs(d["first_step_loss"] - one["first_step_loss"]) < 2e-2 * abs(one["first_step_loss"]), (d["first_step_loss"], one["first_step_loss"])
    assert abs(d["loss"] - one["loss"]) < 3e-2 * abs(one["loss"]), (d["loss"], one["loss"])


def _run_rccl_one_rank(extra, env_extra=None):
    """bench.py under torch.distributed.run with ONE rank and the nccl (= RCCL) backend forced: librccl loads, the communicator
    is created with device_id= on gfx950 under HSA_ENABLE_IPC_MODE_LEGACY=0, and every collective of the N > 1 path executes."""
    env = dict(os.environ, MODA_BENCH_FORCE_NCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MODA_BENCH_ONE_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node=1", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline", "--no-fp32",
           "--no-configs", "--settle", "0"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_rccl_executes_at_world_size_one_render():
    """VERDICT r04 #3a: RCCL itself, on the hardware that exists.  The render bench with its loss all-reduce, barriers and
    max-over-ranks going through an RCCL communicator of one rank; the loss equals the run without a process group."""
    base = ["--rays", "4096", "--samples", "64", "--steps", "3", "--warmup", "1"]
    plain = _run(base, n=1)
    assert plain["collective_backend"] is None
    d = _run_rccl_one_rank(base)
    assert d["collective_backend"] == "nccl" and d["n_ranks_seen"] == 1 and d["n_gpus"] == 1
    assert abs(d["loss"] - plain["loss"]) < 1e-6 * abs(plain["loss"]), (d["loss"], plain["loss"])


def test_rccl_executes_at_world_size_one_training_step():
    """... and the training step: the flat gradient bucket (+ loss sums) all-reduced over RCCL between the two HIP graphs, three
    timed steps; the loss equals the single-graph run without a process group to the atomics' order."""
    args = ["--mode", "train", "--rays", "256", "--samples", "32", "--steps", "3", "--warmup", "1", "--settle-steps", "2"]
    plain = _run(args, n=1)
    assert plain["graph_form"] == "one graph" and plain["collective_backend"] is None
    d = _run_rccl_one_rank(args)
    assert d["collective_backend"] == "nccl" and d["n_ranks_seen"] == 1
    assert d["hip_graph"] is True and d["graph_form"] == "two graphs around one eager all-reduce", d["graph_form"]
    assert d["optimizer_steps"] == plain["optimizer_steps"]
    assert abs(d["loss"] - plain["loss"]) < 2e-3 * abs(plain["loss"]), (d["loss"], plain["loss"])
    assert d["param_checksum_min"] == d["param_checksum_max"]


def test_rccl_all_reduce_inside_the_single_graph_form():
    """VERDICT r05 #6b: MODA_GRAPH_COLLECTIVE=1 -- the whole step INCLUDING the RCCL all-reduce of the gradient bucket as ONE HIP
    graph (thread-local capture mode) -- executed on the hardware that exists: one rank, real RCCL.  Same loss as the two-graph
    form, identical checksums."""
    args = ["--mode", "train", "--rays", "256", "--samples", "32", "--steps", "3", "--warmup", "1", "--settle-steps", "2"]
    two = _run_rccl_one_rank(args)
    d = _run_rccl_one_rank(args, env_extra={"MODA_GRAPH_COLLECTIVE": "1"})
    assert d["collective_backend"] == "nccl" and d["hip_graph"] is True
    assert d["graph_form"] == "one graph with the all-reduce inside", d["graph_form"]
    assert d["optimizer_steps"] == two["optimizer_steps"]
    assert abs(d["loss"] - two["loss"]) < 2e-3 * abs(two["loss"]), (d["loss"], two["loss"])
    assert d["param_checksum_min"] == d["param_checksum_max"]
