"""A few launches of the fused 8 x 256 bf16 kernel at config 2's size (profiling target; MODA_MLP_AGPR selects the form)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import numpy as np, torch
from moda_amd import synth
from moda_amd.bench_support import T, nerf_from_params
torch.set_grad_enabled(False)
N, S = 65536, 256
xyz = torch.from_numpy(np.float32(0.3) * synth.normal(5, "mb/xyz", (4096 * 16, 3))).cuda().repeat(N * S // (4096 * 16), 1).contiguous()
kw = dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False)
m = nerf_from_params(synth.nerf_params(5, "mb/coarse", D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3), **kw)
dirs = T(synth.normal(5, "mb/dir", (N, 91)))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    m.fused(xyz.view(N, S, 3), dir_src=dirs, precision=os.environ.get("MODA_ONE_PREC", "bf16"))
torch.cuda.synchronize()
