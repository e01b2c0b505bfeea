"""GPU test helpers: the scene / workload builders live in moda_amd/bench_support.py (bench.py times the same objects the
tests check); this module re-exports them under the names the tests use and adds the builders only tests need."""
import torch

import moda_amd
from moda_amd import bench_support as _bs
from moda_amd.bench_support import (T, nerf_from_params, make_models, make_opts, rays_to_gpu, TrainHarness,  # noqa: F401
                                    TRAIN_TERMS, TRAIN_WEIGHTS)

DEV = _bs.DEV


def unc_models(seed, B=25):
    """The G19 scene on the GPU: coarse net with the 128-wide appearance code, skin net, uncertainty head
    (moda.py:263-273, 456-464)."""
    from helpers import unc_scene_params
    mp = unc_scene_params(seed)
    models, emb = make_models(seed, B, with_skin=True, perturb_bones=True)
    models["coarse"] = nerf_from_params(mp["coarse"], in_channels_xyz=63, in_channels_dir=27 + 64 + 128, init_beta=0.1)
    unc = moda_amd.NeRFUnc(in_channels_xyz=63, D=8, W=256, out_channels=1, in_channels_dir=32, raw_feat=True, init_beta=1.)
    unc.load_state_dict({k: torch.from_numpy(v) for k, v in mp["nerf_unc"].items()})
    models["nerf_unc"] = unc.to(DEV).eval()
    return models, emb


