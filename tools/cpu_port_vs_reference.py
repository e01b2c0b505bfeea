"""BUILD CONTAINER ONLY (imports /root/reference): the speed of the CPU port that bench.py reports as `cpu_baseline`
(oracle/torch_ref.py, kind "port") against the REFERENCE's own `render_rays` on the same host threads, BASELINE config 1 shapes
(64 samples per ray, 25 bones, eval / no_grad; 1024 rays per call by default so that the pair runs in a minute).

    python tools/cpu_port_vs_reference.py [--rays 1024] [--threads 1,8] [--reps 3]

Writes oracle/port_vs_reference.json (committed): bench.py copies its ratio into `cpu_baseline.port_vs_reference_speed`, so the
record says how far the stated baseline is from the reference itself (the reference's Python cannot travel to the GPU box)."""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")]
ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=1024)
ap.add_argument("--threads", default="1,8")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
import gen_golden as G                                    # imports the reference (tests/golden/_ref_import.py)
from moda_amd import synth
from oracle import torch_ref as tr
sys.path.insert(0, ROOT)
import bench

B, S, N = 25, 64, args.rays
models, emb = G.ref_scene(0, B)
rays_np = synth.make_rays(0, N, B, rays_per_frame=256)
rays = {k: torch.from_numpy(v) for k, v in rays_np.items()}
scene = bench._torch_scene(0, B)
opts = G.make_opts()


def best(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


out = {"rays": N, "samples": S, "bones": B, "torch": torch.__version__, "host_cpus": os.cpu_count(), "threads": {}}
with torch.no_grad():
    for th in [int(t) for t in args.threads.split(",")]:
        torch.set_num_threads(th)
        t_ref = best(lambda: G.rendering.render_rays(models, emb, rays, N_samples=S, chunk=1024 * 32, img_size=512, opts=opts,
                                                     noise_std=0.0), args.reps)
        t_port = best(lambda: tr.render_rays(scene, rays, S), args.reps)
        a = G.rendering.render_rays(models, emb, rays, N_samples=S, chunk=1024 * 32, img_size=512, opts=opts, noise_std=0.0)
        b = tr.render_rays(scene, rays, S)
        err = max(float((a[k] - b[k]).abs().max() / a[k].abs().max()) for k in ("img_coarse", "depth_rnd", "sil_coarse"))
        out["threads"][str(th)] = {"reference_rays_per_s": N / t_ref, "port_rays_per_s": N / t_port,
                                   "port_vs_reference_speed": t_ref / t_port, "max_rel_diff": err}
        print(th, out["threads"][str(th)], flush=True)
out["port_vs_reference_speed"] = min(v["port_vs_reference_speed"] for v in out["threads"].values())
out["note"] = ("measured in the build container (the reference cannot run on the GPU box); < 1 means the port is SLOWER than the "
               "reference, i.e. cpu_baseline.value understates the reference's CPU rate by that factor")
json.dump(out, open(os.path.join(ROOT, "oracle", "port_vs_reference.json"), "w"), indent=1)
print(json.dumps(out))
