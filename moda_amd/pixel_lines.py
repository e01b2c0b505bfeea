"""The per-line pixel files around the path and the gathers that turn a loaded batch into the
observed per-ray signals `inference_deform`'s loss block reads.

On disk (preprocess/img2lines.py:33-110, utils/io.py:380-454), for every forward frame pair (frame t, frame t+dt):

    <root>/<seq>/<dt>_<frame:05d>/rtk.npy     pickled dict {'rtk', 'kaug'}
    <root>/<seq>/<dt>_<frame:05d>/<line:04d>.npy   pickled dict of the image row `line` of both frames:
        img (1,2,3,W)  mask (1,2,W)... -- every array is the full-frame array with its row axis (-2) indexed away

so that a training batch is a set of image ROWS instead of whole frames.  `write_pair` / `read_line` are the two ends of
that format; `set_input`, `obs_to_rays` and `obs_to_rays_line` restate the layout changes moda.set_input
(moda.py:1329-1360) and moda.obs_to_rays[_line] (moda.py:1215-1260) apply before `render_rays` is called.  This module is
file I/O and index selection only (numpy pickles, torch gathers); it performs no arithmetic except the reference's
feature normalisation (a HIP kernel) and mask product."""
import os

import numpy as np
import torch

from . import autograd as A

LINE_KEYS = ('img', 'mask', 'vis2d', 'flow', 'occ', 'dp', 'dp_feat_rsmp')        # img2lines.py:33-42
DFRAMES = (2, 4, 8, 16, 32)                                                       # utils/io.py:420


def dict2pix(dict_array, idy):
    """img2lines.py:33-42: row `idy` of every per-pixel array of a loaded frame pair."""
    return {k: dict_array[k][..., idy, :] for k in LINE_KEYS}


def dict2rtk(dict_array):
    """img2lines.py:44-48"""
    return {'rtk': dict_array['rtk'], 'kaug': dict_array['kaug']}


def pair_dir(root, seqname, dframe, frame):
    """'<root>/<seq>/<dt>_<frame>' with the reference's zero padding (utils/io.py:425, img2lines.py:89-90)."""
    return os.path.join(root, seqname, '%d_%05d' % (int(dframe), int(frame)))


def write_pair(save_dir_t, dict_array, img_size):
    """img2lines.py:97-107: rtk.npy and one pickled dict per image row."""
    os.makedirs(save_dir_t, exist_ok=True)
    np.save(os.path.join(save_dir_t, 'rtk.npy'), dict2rtk(dict_array))
    for idy in range(img_size):
        np.save(os.path.join(save_dir_t, '%04d.npy' % idy), dict2pix(dict_array, idy))


def synthetic_pair(seed, t, dt, img_size, W):
    """A deterministic frame pair in the array layout `tensor2array(batch)` hands to img2lines (batch of one pair):
    per-pixel arrays (1, 2, C, H, W) / (1, 2, H, W) plus the pair's cameras."""
    from . import synth
    name = f"pair/{t}/{dt}/"
    H = img_size
    return {
        'img': synth.uniform(seed, name + "img", (1, 2, 3, H, W)), 'mask': (synth.uniform(seed, name + "mask", (1, 2, H, W)) < 0.6).astype(np.float32),
        'vis2d': (synth.uniform(seed, name + "vis", (1, 2, H, W)) < 0.9).astype(np.float32),
        'flow': synth.normal(seed, name + "flow", (1, 2, 2, H, W)), 'occ': synth.uniform(seed, name + "occ", (1, 2, H, W)),
        'dp': synth.uniform(seed, name + "dp", (1, 2, H, W)), 'dp_feat_rsmp': synth.normal(seed, name + "feat", (1, 2, 16, H, W)),
        'rtk': synth.normal(seed, name + "rtk", (1, 2, 4, 4)), 'kaug': synth.uniform(seed, name + "kaug", (1, 2, 4)),
    }


def write_synthetic_sequence(pixel_dir, seed, n_frames, img_size, W):
    """Every forward pair (t, t + dt) the preprocessing saves for a sequence (img2lines.py:56 dframe list, :79-83)."""
    for t in range(n_frames - 1):
        for dt in (1,) + DFRAMES:
            if dt == 1 or (t % dt == 0 and t + dt <= n_frames - 1):
                write_pair(os.path.join(pixel_dir, '%d_%05d' % (dt, t)), synthetic_pair(seed, t, dt, img_size, W), img_size)


def write_synthetic_cameras(cam_dir, seed, n_frames, skip=()):
    """One 4x4 camera text file per frame, '<frame:05d>.txt' (the `Cameras/` twin of `JPEGImages/`, utils/io.py:396);
    frames in `skip` get none, so that pairs touching them take LineDataset's default camera (:438-444)."""
    from . import synth
    os.makedirs(cam_dir, exist_ok=True)
    for i in range(n_frames):
        if i not in skip:
            np.savetxt(os.path.join(cam_dir, '%05d.txt' % i), synth.normal(seed, f"cam/{i}", (4, 4)).astype(np.float64))


def read_line(save_dir_t, idy):
    return np.load(os.path.join(save_dir_t, '%04d.npy' % int(idy)), allow_pickle=True).item()


def default_camera():
    """The camera LineDataset substitutes when a frame has no camera file (utils/io.py:438-444)."""
    rtk = np.zeros((4, 4))
    rtk[:3, :3] = np.eye(3)
    rtk[:3, 3] = np.asarray([0, 0, 10])
    rtk[3, :] = np.asarray([512, 512, 256, 256])
    return rtk


class LineDataset(torch.utils.data.Dataset):
    """utils/io.py:380-454: item `index` = image row index % img_size of frame index // img_size, paired with frame
    t + dframe for a dframe drawn from those the preprocessing saved."""

    def __init__(self, pixel_dir, n_frames, img_size, rtklist=None, dataid=0, rng=None):
        self.pixel_dir = pixel_dir
        self.n_frames = n_frames
        self.img_size = img_size
        self.num_lines = (n_frames - 1) * img_size          # the last frame has no forward pair (:390)
        self.rtklist = rtklist
        self.dataid = dataid
        self.rng = rng or np.random

    def __len__(self):
        return self.num_lines

    def dframe_choices(self, idt):
        max_id = self.n_frames - 1
        return [1] + [i for i in DFRAMES if idt % i == 0 and idt + i <= max_id]   # :420-423

    def __getitem__(self, index):
        idt, idy = index // self.img_size, index % self.img_size
        dframe = int(self.rng.choice(self.dframe_choices(idt)))
        d = os.path.join(self.pixel_dir, '%d_%05d' % (dframe, idt))
        elem = read_line(d, idy)
        idtn = idt + dframe
        try:
            rtk = np.stack([np.loadtxt(self.rtklist[idt]), np.loadtxt(self.rtklist[idtn])])
        except Exception:
            rtk = np.stack([default_camera(), default_camera()])
        kaug = np.load(os.path.join(d, 'rtk.npy'), allow_pickle=True).item()['kaug']
        elem['rtk'] = rtk[None]
        elem['kaug'] = kaug
        elem['dataid'] = np.stack([self.dataid, self.dataid])[None]
        elem['frameid'] = np.stack([idt, idtn])[None]
        elem['lineid'] = np.stack([idy, idy])[None]
        return elem


def set_input(batch, data_offset, img_size, device='cuda'):
    """moda.set_input (moda.py:1329-1360): a collated batch (bs pairs) -> pair-major device tensors
    (first all frames t, then all frames t+dt), pixels flattened on axis 2."""
    b = {k: torch.as_tensor(v).float() for k, v in batch.items()}
    bs = b['dataid'].shape[0]
    pm = lambda t, c: t.view(bs, 2, c, -1).permute(1, 0, 2, 3).reshape(bs * 2, c, -1, 1).to(device)
    out = {'imgs': pm(b['img'], 3), 'masks': pm(b['mask'], 1), 'vis2d': pm(b['vis2d'], 1), 'flow': pm(b['flow'], 2),
           'occ': pm(b['occ'], 1), 'dps': pm(b['dp'], 1)}
    f = pm(b['dp_feat_rsmp'], 16)                                                  # F.normalize(., 2, 1) (:1345)
    P = f.shape[2]
    out['dp_feats'] = A.NormalizeFn.apply(f[..., 0].permute(0, 2, 1).reshape(-1, 16)).view(2 * bs, P, 16).permute(0, 2, 1)[..., None]
    out['rtk'] = b['rtk'].view(bs, -1, 4, 4).permute(1, 0, 2, 3).reshape(-1, 4, 4).to(device)
    out['kaug'] = b['kaug'].view(bs, -1, 4).permute(1, 0, 2).reshape(-1, 4).to(device)
    fl = lambda t: t.view(bs, -1).permute(1, 0).reshape(-1)
    frameid, dataid = fl(b['frameid']).cpu(), fl(b['dataid']).cpu()
    out['lineid'] = fl(b['lineid']).to(device) if 'lineid' in b else None
    off = torch.as_tensor(np.asarray(data_offset)).float()[dataid.long()]
    out['frameid_sub'] = frameid.clone()
    out['dataid'] = dataid
    out['embedid'] = frameid + off
    out['frameid'] = frameid + off
    if out['lineid'] is not None:
        out['errid'] = out['frameid'] * img_size + out['lineid'].cpu()
    out['masks'] = ((out['masks'] * out['vis2d']) > 0).float()                    # :1358-1359
    return out


def obs_to_rays(rays, rand_inds, imgs, masks, vis2d, flow, occ, dp_feats=None):
    """moda.obs_to_rays (moda.py:1237-1260): frame tensors (bs, C, P[, 1]) and pixel indices (bs, ns) -> (bs, ns, C)."""
    idx = rand_inds.long().to(imgs.device)
    pick = lambda t, c: torch.gather(t.reshape(t.shape[0], c, -1), 2, idx[:, None].expand(-1, c, -1)).permute(0, 2, 1)
    rays['img_at_samp'] = pick(imgs, 3)
    rays['sil_at_samp'] = pick(masks, 1)
    rays['vis_at_samp'] = pick(vis2d, 1)
    rays['flo_at_samp'] = pick(flow, 2)
    rays['cfd_at_samp'] = pick(occ, 1)
    if dp_feats is not None:
        rays['feats_at_samp'] = pick(dp_feats, 16)
    return rays


def obs_to_rays_line(rays, rand_inds, imgs, masks, vis2d, flow, occ, dp_feats, batch_map):
    """moda.obs_to_rays_line (moda.py:1215-1235): one pixel per entry -- rand_inds (R, 1) columns of row
    batch_map (R,) -> (R, 1, C)."""
    idx = rand_inds.long().to(imgs.device)
    bm = batch_map.long().to(imgs.device)
    pick = lambda t, c: torch.gather(t[bm][..., 0], 2, idx[:, None].expand(-1, c, -1))[:, None][..., 0]
    rays['img_at_samp'] = pick(imgs, 3)
    rays['sil_at_samp'] = pick(masks, 1)
    rays['vis_at_samp'] = pick(vis2d, 1)
    rays['flo_at_samp'] = pick(flow, 2)
    rays['cfd_at_samp'] = pick(occ, 1)
    if dp_feats is not None:
        rays['feats_at_samp'] = pick(dp_feats, 16)
    return rays
