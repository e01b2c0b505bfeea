import sys, os, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
variant = sys.argv[1] if len(sys.argv) > 1 else ""
if variant:
    os.environ["MODA_HIPCC_FLAGS"] = variant
    from moda_amd import build
    build.build(force=True, verbose=False)
import moda_amd
from moda_amd import synth
from oracle import moda_oracle as orc
from gpu_helpers import T, nerf_from_params
torch.set_grad_enabled(False)
np.set_printoptions(linewidth=250, precision=4, suppress=True)
def run(W, D, n_out, M, prec, show=True):
    kw = dict(D=D, W=W, in_channels_xyz=63, in_channels_dir=0, out_channels=n_out, raw_feat=True)
    pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
    p = synth.nerf_params(13, f"dbg/{W}/{D}/{n_out}", **pk)
    m = nerf_from_params(p, **kw)
    xyz = np.float32(0.35) * synth.normal(13, "dbg/xyz", (M, 3))
    out = m.fused(T(xyz), precision=prec).cpu().numpy()
    out2 = m.fused(T(xyz), precision=prec).cpu().numpy()
    x = orc.embedding(xyz, 10, 10.0)
    ref = orc.nerf_forward(p, x, D=D, W=W, in_channels_xyz=63, in_channels_dir=0, raw_feat=True,
                           round_fn=orc.bf16_round if prec == "bf16" else None)
    err = np.abs(out - ref); sc = np.abs(ref).max()
    e = err.max(1)/sc
    bad = np.nonzero(e > 0.008)[0]
    print(f"[{variant}] W={W} D={D} n_out={n_out} M={M} {prec}: rel {err.max()/sc:.3e} repeat-diff {np.abs(out-out2).max():.2e} bad samples: {len(bad)}", 
          "blocks(32):", sorted(set((bad//32).tolist()))[:20], "cols:", sorted(set((bad%32).tolist())))
for (W, D, n_out, M) in [(128,5,16,256),(128,5,16,64),(128,5,16,1024),(128,7,16,256),(128,8,16,256),(256,5,16,256),(64,5,16,256)]:
    run(W, D, n_out, M, "bf16")
