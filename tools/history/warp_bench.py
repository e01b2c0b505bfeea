"""Micro-benchmark of the fused skinning + DQS warp kernel at the bench shape (65536 rays x 256 samples, 25 bones).
usage: python tools/warp_bench.py ["<hipcc flags>"]"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
os.environ["MODA_HIPCC_FLAGS"] = sys.argv[1] if len(sys.argv) > 1 else ""
from moda_amd import build
build.build(force=True, verbose=False)
from moda_amd import synth, geom_utils as G
from gpu_helpers import T
torch.set_grad_enabled(False)
N, S, B = 65536, 256, 25
bones = T(synth.make_models(0, B=B, with_skin=False, perturb_bones=True)["bones_rst"])
rts = T(synth.frame_dual_quats(0, "wb/rts", N // 256, B)).repeat_interleave(256, 0).contiguous()
xyz = (0.2 * torch.randn(N, S, 3, device="cuda")).contiguous()
dskin = torch.randn(N, B, S, device="cuda")
aux = T(np.asarray([0.0, 10], np.float32))
bd = G.bone_transform(bones, rts, True, is_vec=True)
def run():
    return G.warp(bd, rts, xyz, dskin, aux, backward=True, dskin_bns=True)
run(); torch.cuda.synchronize()
ts = []
for _ in range(7):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); run(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
gb = (N * S * (12 + 12 + 4 * B)) / 1e9
print(f"[{os.environ['MODA_HIPCC_FLAGS'] or 'default'}] warp (incl. prep kernels): {np.median(ts):.3f} ms (min {min(ts):.3f}) = {gb/np.median(ts):.2f} TB/s algorithmic")
