"""GPU probe (development): the fp16 mode against the fp16-rounding oracle, the fp32 oracle and the reference goldens, its speed
beside the bf16 mode on config 2, the overflow flag, and fp16 subnormal weights through the MFMA."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import moda_amd
from moda_amd import synth, overflow
from oracle import moda_oracle as orc
from helpers import E2E_CASES, elem_err, golden, rel_err
from gpu_helpers import T, make_models, make_opts, rays_to_gpu
import test_gpu_parity as tp

np_ = lambda t: t.detach().cpu().numpy()
torch.set_grad_enabled(False)
for name in ("coarse", "skin", "feat", "vis"):
    e1 = tp._fused_vs_oracle(name, M=64 * 16, n_rows=64, precision="fp16", round_fn=orc.f16_round, tol=1.0)
    e2 = tp._fused_vs_oracle(name, M=64 * 16, n_rows=64, precision="fp16", round_fn=None, tol=1.0)
    e3 = tp._fused_vs_oracle(name, M=64 * 16, n_rows=64, precision="bf16", round_fn=None, tol=1.0)
    print(f"fp16 {name}: vs f16-oracle {e1:.2e}, vs fp32-oracle {e2:.2e} (bf16 mode vs fp32-oracle {e3:.2e})", flush=True)
overflow.check()
for name in E2E_CASES:
    for prec in ("fp16", "bf16"):
        res, g = tp.run_hip_case(name, precision=prec)
        worst = (0.0, 0.0, "")
        wr = (0.0, "")
        for k in [k for k in g if not k.startswith("rng")]:
            err, ee = rel_err(np_(res[k]), g[k]), elem_err(np_(res[k]), g[k])
            worst = max(worst, (ee, err, k)); wr = max(wr, (err, k))
        if prec == "fp16" and worst[0] >= 1:
            for k in [k for k in g if not k.startswith("rng")]:
                print(f"      {k}: rel {rel_err(np_(res[k]), g[k]):.2e} elem {elem_err(np_(res[k]), g[k]):.3f}")
        print(f"g7 {name} ({prec}): worst elem {worst[0]:.3f} on {worst[2]} (rel {worst[1]:.2e}); worst rel {wr[0]:.2e} on {wr[1]}", flush=True)
overflow.check()
g = golden("g8_cfg1")
models, emb = make_models(0, 25)
rays = rays_to_gpu(synth.make_rays(0, 4096, 25, rays_per_frame=256))
for prec in ("fp16", "bf16", "bf16x3"):
    moda_amd.set_precision(prec)
    res = moda_amd.render_rays(models, emb, rays, N_samples=64, noise_std=0.0, opts=make_opts(), img_size=512)
    idx = g["ray_index"]
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "xyz_camera_vis", "xyz_canonical_vis", "frame_cyc_dis"):
        a = np_(res[k])
        print(f"g8 {k} ({prec}): rel {rel_err(a[idx], g[k + '_rays']):.2e}, elem {elem_err(a[idx], g[k + '_rays']):.3f}", flush=True)
overflow.check()
# cfg2 speed + deviation from the bf16x3 / fp32 modes
N, S = 65536, 256
models, emb = make_models(0, 25)
rays = rays_to_gpu(synth.make_rays(0, N, 25, rays_per_frame=256))
outs = {}
for prec in ("bf16", "fp16"):
    moda_amd.set_precision(prec)
    for _ in range(3):
        res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10):
        res = moda_amd.render_rays(models, emb, rays, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    outs[prec] = {k: v.clone() for k, v in res.items() if torch.is_tensor(v)}
    print(f"cfg2 {prec}: {dt*1e3:.2f} ms/step = {N/dt/1e6:.3f} M rays/s", flush=True)
moda_amd.set_precision("bf16x3")
sub = {k: (v[:16384 // (256 if v.shape[0] != N else 1)] if torch.is_tensor(v) else v) for k, v in rays.items()}
sub = {k: (v[:16384] if torch.is_tensor(v) and v.shape[0] == N else v) for k, v in rays.items()}
ref = moda_amd.render_rays(models, emb, sub, N_samples=S, noise_std=0.0, opts=make_opts(), img_size=512)
for prec in ("bf16", "fp16"):
    for k in ("img_coarse", "depth_rnd", "sil_coarse", "xyz_canonical_vis", "frame_cyc_dis"):
        a, b = np_(outs[prec][k][:16384]), np_(ref[k])
        print(f"cfg2 {prec} vs bf16x3 {k}: rel {rel_err(a, b):.2e} elem {elem_err(a, b):.3f}", flush=True)
overflow.check()
# overflow flag: a network whose first layer is scaled by 1e6
moda_amd.set_precision("fp16")
kw, p, m = tp._nerf_case("skin", seed=13, tag="fused/")
xyz = T(np.float32(0.35) * synth.normal(13, "ovf/xyz", (4096, 3)))
code = T(synth.normal(13, "ovf/code", (1, kw["in_channels_xyz"] - 63)))
out = m.fused(xyz, code=code)
torch.cuda.synchronize(); print("flag after a normal launch:", overflow.tripped())
m.xyz_encoding_2[0].weight.data.mul_(3e5)
out = m.fused(xyz, code=code)
torch.cuda.synchronize(); print("flag after a 3e5-scaled layer:", overflow.tripped(), "finite outputs:", bool(torch.isfinite(out).all()))
try:
    overflow.check(); print("NOT RAISED")
except overflow.Fp16Overflow as e:
    print("raised:", str(e)[:60])
m.xyz_encoding_2[0].weight.data.mul_(1e3)         # weights beyond 65504 themselves
out = m.fused(xyz, code=code)
torch.cuda.synchronize(); print("flag after weights > fp16 max:", overflow.tripped()); overflow.reset()
# subnormal fp16 weights: a layer scaled by 1e-4 (weights ~ 6e-6, fp16 subnormal), next layer scaled back by 1e4
kw, p, m = tp._nerf_case("skin", seed=13, tag="fused/")
o0 = m.fused(xyz, code=code, precision="fp32")
m.xyz_encoding_3[0].weight.data.mul_(1e-3); m.xyz_encoding_3[0].bias.data.mul_(1e-3)
m.xyz_encoding_4[0].weight.data.mul_(1e3)
o1 = m.fused(xyz, code=code, precision="fp32")
o2 = m.fused(xyz, code=code, precision="fp16")
print("subnormal-weight layer: fp32 self-consistency", rel_err(np_(o1), np_(o0)), " fp16 vs fp32:", rel_err(np_(o2), np_(o1)))
overflow.check()
