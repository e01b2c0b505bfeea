"""Time one training step of the rendering path (forward + backward through the HIP autograd Functions + AdamW)
at the reference's recipe size (scripts/template.sh: 2048 rays x 128 samples per GPU) -- SURVEY.md config 4 shape."""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import moda_amd
from moda_amd import synth
from gpu_helpers import make_models, make_opts, rays_to_gpu

N, S, B = int(os.environ.get("RAYS", 2048)), int(os.environ.get("SAMPLES", 128)), 25
models, emb = make_models(0, B)
for m in models.values():
    if isinstance(m, torch.nn.Module):
        m.train()
models["bones_rst"] = torch.nn.Parameter(models["bones_rst"].clone())
models["skin_aux"] = torch.nn.Parameter(models["skin_aux"].clone())
rays = rays_to_gpu(synth.make_rays(0, N, B, rays_per_frame=4))
for k in ("bone_rts", "time_embedded", "env_code", "rays_o", "rays_d"):
    rays[k].requires_grad_(True)
target = torch.from_numpy(synth.uniform(1, "t", (N, 3))).cuda()
params = [p for m in models.values() if isinstance(m, torch.nn.Module) for p in m.parameters()] + [models["bones_rst"], models["skin_aux"]]
opt = torch.optim.AdamW(params, lr=5e-4)

def step():
    opt.zero_grad(set_to_none=True)
    res = moda_amd.render_rays(models, emb, rays, N_samples=S, perturb=1.0, noise_std=0.0, opts=make_opts(), img_size=512)
    loss = (res["img_coarse"] - target).pow(2).mean() + 0.1 * (res["sil_coarse"] - 1).pow(2).mean() + 0.05 * res["frame_cyc_dis"].mean()
    loss.backward()
    opt.step()
    return loss

for _ in range(3):
    l = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
flop = 3 * 2 * (601600 + 2 * 47840) * N * S
print(f"train step {N}x{S}: {dt*1e3:.2f} ms  ({N/dt:.0f} rays/s, {flop/dt/1e12:.1f} TFLOP/s fwd+bwd algorithmic, loss {float(l):.4f})")
