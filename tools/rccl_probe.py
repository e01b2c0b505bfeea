"""World-size-1 RCCL probe on the one GPU a gpurun box has: does librccl load, does the device_id= init work on gfx950 under
HSA_ENABLE_IPC_MODE_LEGACY=0, does all_reduce run, and can a collective be captured into a HIP graph (torch.cuda.graph)?
    python tools/rccl_probe.py"""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
t0 = time.time()
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
dist.all_reduce(x)
torch.cuda.synchronize()
print(f"rccl init + first all_reduce: {time.time() - t0:.2f} s; sum ok: {bool((x == torch.arange(1 << 20, device='cuda')).all())}", flush=True)
try:
    print("nccl version:", torch.cuda.nccl.version())
except Exception as e:
    print("nccl version: ?", e)
# eager latency of small / bucket-sized all-reduces
for n in (2, 2_900_000):
    y = torch.ones(n, device="cuda")
    for _ in range(5):
        dist.all_reduce(y)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        dist.all_reduce(y)
    torch.cuda.synchronize()
    print(f"all_reduce of {n} floats: {(time.perf_counter() - t) / 50 * 1e6:.1f} us eager", flush=True)
# capture: the process group's watchdog thread polls its work events (hipEventQuery); under the default GLOBAL capture mode
# that call is illegal while ANY thread captures and takes the process down -- thread-local mode confines the check to this thread
import time as _t
for mode in ("thread_local", "relaxed"):
    try:
        y = torch.ones(2_900_000, device="cuda")
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            for _ in range(3):
                dist.all_reduce(y)
        torch.cuda.current_stream().wait_stream(s_)
        torch.cuda.synchronize()
        _t.sleep(0.5)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=mode):
            z = y * 2
            dist.all_reduce(z)
            w = z + 1
        torch.cuda.synchronize()
        y.fill_(3.0)
        g.replay()
        torch.cuda.synchronize()
        print(f"graph capture of all_reduce [{mode}]: ok; replay result", float(w[0]), "(expected 7.0)", flush=True)
        t = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        print(f"graph replay (mul + all_reduce + add): {(time.perf_counter() - t) / 50 * 1e6:.1f} us", flush=True)
        break
    except Exception as e:
        print(f"graph capture of all_reduce [{mode}] FAILED:", type(e).__name__, str(e).splitlines()[0], flush=True)
dist.destroy_process_group()
print("done")
