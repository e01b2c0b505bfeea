"""On-disk formats either side of the path: MoDA's checkpoints `params_<label>.pth` (the
trainer's `model.state_dict()`, nnutils/train_utils.py:292-306) and `vars_<label>.npy` (`latest_vars`, a pickled dict
with `obj_bound` among others, :298-304), read straight into the objects `render_rays` takes.

The state-dict key names are the reference model's attribute names (nnutils/moda.py:186-465): `nerf_coarse.*`,
`nerf_skin.*`, `nerf_feat.*`, `nerf_vis.*`, `nerf_unc.*` (NeRF modules, nerf.py:109-140), `bones`, `skin_aux`,
`rest_pose_code.weight`, `pose_code.basis_mlp.*`, `env_code.basis_mlp.*`, `nerf_body_rts.1.*` (DQ_RTHead; `.0` is
the shared pose code), `vid_code.weight`, `alpha`, `near_far`.  tests/golden/g20_checkpoint.npz holds that key -> shape
map as the reference's own classes produce it.  Network shapes are read off the tensors, so no option file is needed."""
import numpy as np
import torch

from .feeders import DQ_RTHead, FrameCode
from .nerf import Embedding, NeRF, NeRFUnc


def rm_module_prefix(states, prefix='module'):
    """train_utils.py:308-316: strip a DistributedDataParallel 'module.' prefix."""
    out = {}
    for k, v in states.items():
        out[k[len(prefix) + 1:] if k.startswith(prefix + '.') else k] = v
    return out


def load_params(path, rm_prefix=True):
    states = torch.load(path, map_location='cpu')
    return rm_module_prefix(states) if rm_prefix else states


def load_vars(path):
    """vars_<label>.npy -> dict; `obj_bound` broadcast to 3 components as train_utils.py:368-370 does."""
    v = np.load(path, allow_pickle=True)[()]
    if 'obj_bound' in v and np.size(v['obj_bound']) == 1:
        v['obj_bound'] = v['obj_bound'] * np.ones(3)
    return v


def _sub(states, prefix):
    return {k[len(prefix) + 1:]: v for k, v in states.items() if k.startswith(prefix + '.')}


def _nerf_from_states(sd, cls=NeRF, **extra):
    """Build a NeRF whose constructor arguments are read off the tensors' shapes (nerf.py:84-140)."""
    D = sum(1 for k in sd if k.startswith('xyz_encoding_') and k.endswith('.0.weight') and 'final' not in k)
    W, in_xyz = sd['xyz_encoding_1.0.weight'].shape
    in_dir = sd['dir_encoding.0.weight'].shape[1] - W
    n_out = sd['rgb.0.weight'].shape[0]
    m = cls(D=D, W=W, in_channels_xyz=in_xyz, in_channels_dir=in_dir, out_channels=n_out, **extra)
    m.load_state_dict(sd)
    return m


def build_models(states, device='cuda', data_offset=None, num_freqs=10):
    """states (a MoDA state dict) -> (models, embeddings, extras): `models` / `embeddings` are render_rays' first two
    arguments (moda.py:277-349, 444-465); `extras` holds the per-frame feeders found in the checkpoint
    (`pose_code`, `env_code`, `nerf_body_rts`; FrameCode needs `data_offset`, the video boundaries of the dataset)."""
    s = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in states.items()}
    alpha = float(s['alpha'].reshape(-1)[0]) if 'alpha' in s else float(num_freqs)
    embeddings = {'xyz': Embedding(3, num_freqs, alpha=alpha), 'dir': Embedding(3, 4, alpha=alpha)}
    models = {'coarse': _nerf_from_states(_sub(s, 'nerf_coarse')).to(device).eval()}
    for key, attr, raw in (('nerf_skin', 'nerf_skin', True), ('nerf_feat', 'nerf_feat', True), ('nerf_vis', 'nerf_vis', True),
                           ('nerf_dis', 'nerf_dis', True)):
        sd = _sub(s, attr)
        if sd:
            models[key] = _nerf_from_states(sd, raw_feat=raw).to(device).eval()
    sd = _sub(s, 'nerf_unc')
    if sd:
        models['nerf_unc'] = _nerf_from_states(sd, cls=NeRFUnc, raw_feat=True).to(device).eval()
    if 'bones' in s:
        models['bones'] = torch.nn.Parameter(s['bones'].to(device))
        models['bones_rst'] = s['bones'].to(device).clone()       # update_delta_rts replaces it per step (moda.py:1267-1268)
        models['skin_aux'] = s['skin_aux'].to(device)
        if 'rest_pose_code.weight' in s:
            rpc = torch.nn.Embedding(*s['rest_pose_code.weight'].shape)
            rpc.weight.data = s['rest_pose_code.weight']
            models['rest_pose_code'] = rpc.to(device)
    extras = {}
    if 'near_far' in s:
        extras['near_far'] = s['near_far'].to(device)
    for name in ('pose_code', 'env_code'):
        w = s.get(name + '.basis_mlp.weight')
        if w is not None:
            if data_offset is None:
                raise ValueError(f"{name} is a FrameCode: pass data_offset (the dataset's video boundaries)")
            n_vids = len(data_offset) - 1
            if n_vids < 1 or w.shape[1] % n_vids or (w.shape[1] // n_vids) % 2 != 1:       # in = n_vids (1 + 2F), nerf.py:359-361
                raise ValueError(f"{name}.basis_mlp.weight has {w.shape[1]} input columns, which is not n_vids * (1 + 2F) "
                                 f"for the {n_vids} videos of data_offset={list(data_offset)}")
            fc = FrameCode((w.shape[1] // n_vids - 1) // 2, w.shape[0], np.asarray(data_offset))
            fc.load_state_dict(_sub(s, name))
            extras[name] = fc.to(device)
    if 'vid_code.weight' in s:                                  # per-video code of the uncertainty head (moda.py:459-460)
        vc = torch.nn.Embedding(*s['vid_code.weight'].shape)
        vc.weight.data = s['vid_code.weight']
        extras['vid_code'] = vc.to(device)
    sd = _sub(s, 'nerf_body_rts.1')
    if sd and 'pose_code' in extras:
        W, in_xyz = sd['xyz_encoding_1.0.weight'].shape
        D = sum(1 for k in sd if k.startswith('xyz_encoding_') and k.endswith('.0.weight') and 'final' not in k)
        head = DQ_RTHead(use_quat=True, D=D, W=W, in_channels_xyz=in_xyz, in_channels_dir=0,
                         out_channels=sd['rgb.0.weight'].shape[0], raw_feat=True)
        head.load_state_dict(sd)
        extras['nerf_body_rts'] = torch.nn.Sequential(extras['pose_code'], head.to(device).eval())
    return models, embeddings, extras
