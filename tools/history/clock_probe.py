"""Is the 8x256 bf16 MLP kernel limited by the clock the chip holds under load (MI355X_MICROARCH.md, DVFS give-back)?
Same instruction stream on random vs all-zero weights and inputs: a large gap means power / clock, not issue, sets the pace."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from moda_amd import synth
from gpu_helpers import T, nerf_from_params
torch.set_grad_enabled(False)
N, S = 65536, 256
kw = dict(D=8, W=256, in_channels_xyz=63, in_channels_dir=91, out_channels=3, raw_feat=False)
pk = {k: kw[k] for k in ("D", "W", "in_channels_xyz", "in_channels_dir", "out_channels")}
p = synth.nerf_params(5, "mb/coarse", **pk)
def timeit(m, xyz, dirs):
    f = lambda: m.fused(xyz, dir_src=dirs, precision="bf16")
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts))
xyz = (0.3 * torch.randn(N * S, 3, device="cuda")).contiguous()
dirs = T(synth.normal(5, "mb/dir", (N, 91)))
m = nerf_from_params(p, **kw)
t_rand = timeit(m, xyz, dirs)
mz = nerf_from_params({k: np.zeros_like(v) for k, v in p.items()}, **kw)
t_zero = timeit(mz, torch.zeros_like(xyz), torch.zeros_like(dirs))
t_rand2 = timeit(m, xyz, dirs)
print(f"random data {t_rand:.3f} ms ({2*601600*N*S/t_rand/1e9:.0f} TFLOP/s) | all zeros {t_zero:.3f} ms ({2*601600*N*S/t_zero/1e9:.0f} TFLOP/s) | random again {t_rand2:.3f} ms  -> zeros are {100*(t_rand/t_zero-1):.1f} % faster")
