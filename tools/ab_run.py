"""Interleaved A/B timing of library variants on ONE box (boxes of the pool differ by +-4 %):
  python tools/ab_run.py [--precision bf16|fp16|...] [--rounds 3] [--steps 40] default ab/<name>.so default:VAR=value ...
Every (round, variant) is a fresh process with MODA_LIB_PATH set; prints ms per render_rays step (config 2) and the event-timed
kernels of the step."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
import numpy as np, torch
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import moda_amd
from moda_amd import synth, _lib
from gpu_helpers import make_models, make_opts, rays_to_gpu
torch.set_grad_enabled(False)
moda_amd.set_precision(%(prec)r)
models, emb = make_models(0, 25)
rays = rays_to_gpu(synth.make_rays(1000, 65536, 25, rays_per_frame=256))
kw = dict(N_samples=256, noise_std=0.0, opts=make_opts(), img_size=512)
t0 = time.time()
while time.time() - t0 < 1.5:
    moda_amd.render_rays(models, emb, rays, **kw); torch.cuda.synchronize()
_lib.PROFILE = {}
torch.cuda.synchronize(); t0 = time.time()
for _ in range(%(steps)d):
    moda_amd.render_rays(models, emb, rays, **kw)
torch.cuda.synchronize(); dt = (time.time() - t0) / %(steps)d
prof = _lib.PROFILE
print(json.dumps({"ms": dt * 1e3, **{t: float(np.mean([s.elapsed_time(e) for s, e, _ in ev])) for t, ev in prof.items()}}))
'''
args = sys.argv[1:]
prec, rounds, steps = "bf16", 3, 40
while args and args[0].startswith("--"):
    k, v = args[0], args[1]; args = args[2:]
    if k == "--precision": prec = v
    elif k == "--rounds": rounds = int(v)
    elif k == "--steps": steps = int(v)
res = {v: [] for v in args}
for r in range(rounds):
    for v in args:
        env = dict(os.environ)
        lib, *sets = v.split(":")                 # "<default | ab/name.so>[:VAR=value[:VAR=value]]"
        if lib != "default":
            env["MODA_LIB_PATH"] = os.path.join(ROOT, "moda_amd", "lib", lib)
        else:
            env.pop("MODA_LIB_PATH", None)
        for kv in sets:
            k_, v_ = kv.split("=", 1)
            env[k_] = v_
        p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, prec=prec, steps=steps)], env=env, capture_output=True, text=True)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception:
            print(v, "FAILED", p.stderr[-400:]); continue
        res[v].append(d)
        print(f"round {r} [{v}] " + " ".join(f"{k}={x:.3f}" for k, x in d.items()), flush=True)
for v, ds in res.items():
    if ds:
        print(f"== {v}: " + " ".join(f"{k}={sum(d[k] for d in ds)/len(ds):.3f}" for k in ds[0]))
