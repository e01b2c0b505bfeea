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


def _run(extra, n=2):
    env = dict(os.environ, MODA_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--no-cpu-baseline", "--no-fp32",
           "--no-configs", "--settle", "0"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]               # rank 0 alone prints, once
    return json.loads(lines[0])


def test_two_ranks_weak_and_strong_render():
    one = _run(["--rays", "4096", "--samples", "64", "--steps", "3", "--warmup", "1"], n=1)
    weak = _run(["--rays", "4096", "--samples", "64", "--steps", "3", "--warmup", "1"])
    strong = _run(["--rays", "4096", "--samples", "64", "--steps", "3", "--warmup", "1", "--scaling", "strong"])
    assert weak["n_gpus"] == strong["n_gpus"] == 2 and weak["n_ranks_seen"] == strong["n_ranks_seen"] == 2
    assert weak["scaling"] == "weak" and strong["scaling"] == "strong"
    assert weak["config"]["rays_per_step"] == 8192 and weak["config"]["rays_per_gpu"] == 4096
    assert strong["config"]["rays_per_step"] == 4096 and strong["config"]["rays_per_gpu"] == 2048
    # strong scaling renders the SAME 4096 rays as the single rank: the all-reduced loss is the single-rank loss
    assert abs(strong["loss"] - one["loss"]) < 1e-6 * abs(one["loss"]), (strong["loss"], one["loss"])
    # weak scaling: rank 0 renders the single rank's rays, rank 1 its own; the loss is the mean over both sets
    assert weak["loss"] != one["loss"] and 0.1 < weak["loss"] < 0.5
    for d in (weak, strong):
        assert d["value"] > 0 and d["unit"] == "rays/s" and "roofline" in d


def test_two_ranks_training_step_exchanges_gradients():
    d = _run(["--mode", "train", "--rays", "256", "--samples", "32", "--steps", "2", "--warmup", "1", "--settle-steps", "2"])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["hip_graph"] is False
    assert d["optimizer_steps"] == 5 and d["loss"] == d["loss"] and 0.5 < d["loss"] < 5
