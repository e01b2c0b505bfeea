"""CPU: the per-line pixel file format (preprocess/img2lines.py:33-110, utils/io.py:380-454) written and read back, and
the observed-signal gathers of moda.obs_to_rays[_line] (moda.py:1215-1260) against direct indexing."""
import os

import numpy as np
import torch

from moda_amd import pixel_lines as PL


def frame_pair(rng, H):
    """A loaded frame pair as tensor2array(batch) hands it to img2lines (batch of 1, 2 frames, H x H pixels)."""
    a = lambda *s: rng.standard_normal(s).astype(np.float32)
    return {'img': a(1, 2, 3, H, H), 'mask': a(1, 2, H, H), 'vis2d': a(1, 2, H, H), 'flow': a(1, 2, 2, H, H),
            'occ': a(1, 2, H, H), 'dp': a(1, 2, H, H), 'dp_feat_rsmp': a(1, 2, 16, H, H),
            'rtk': a(1, 2, 4, 4), 'kaug': a(1, 2, 4)}


def test_line_files_round_trip(tmp_path):
    H, n_frames = 8, 5
    rng = np.random.default_rng(0)
    root = str(tmp_path / "Pixels" / "seq")
    pairs = {}
    for idt in range(n_frames - 1):
        for dt in [1] + [d for d in PL.DFRAMES if idt % d == 0 and idt + d <= n_frames - 1]:
            pairs[(dt, idt)] = frame_pair(rng, H)
            PL.write_pair(os.path.join(root, '%d_%05d' % (dt, idt)), pairs[(dt, idt)], H)
    assert sorted(os.listdir(os.path.join(root, '1_00000'))) == ['%04d.npy' % i for i in range(H)] + ['rtk.npy']
    ds = PL.LineDataset(root, n_frames, H, rtklist=None, dataid=2, rng=np.random.default_rng(1))
    assert len(ds) == (n_frames - 1) * H
    seen_dt = set()
    for index in range(len(ds)):
        e = ds[index]
        idt, idy = index // H, index % H
        dt = int(e['frameid'][0, 1] - e['frameid'][0, 0])
        seen_dt.add(dt)
        assert dt in ds.dframe_choices(idt) and e['frameid'][0, 0] == idt
        assert (e['lineid'] == idy).all() and (e['dataid'] == 2).all()
        src = pairs[(dt, idt)]
        for k in PL.LINE_KEYS:
            assert e[k].shape == src[k].shape[:-2] + (H,)
            assert np.array_equal(e[k], src[k][..., idy, :])
        assert np.array_equal(e['kaug'], src['kaug'])
        assert e['rtk'].shape == (1, 2, 4, 4) and np.array_equal(e['rtk'][0, 0], PL.default_camera())
    assert seen_dt == {1, 2, 4}


def test_cameras_are_read_from_text_files(tmp_path):
    H = 4
    rng = np.random.default_rng(2)
    root = str(tmp_path / "seq")
    PL.write_pair(os.path.join(root, '1_00000'), frame_pair(rng, H), H)
    cams = [rng.standard_normal((4, 4)) for _ in range(2)]
    paths = []
    for i, c in enumerate(cams):
        paths.append(str(tmp_path / f"cam-{i:05d}.txt"))
        np.savetxt(paths[-1], c)
    e = PL.LineDataset(root, 2, H, rtklist=paths)[3]
    assert np.allclose(e['rtk'][0], np.stack(cams))


def test_obs_to_rays_matches_direct_indexing():
    rng = np.random.default_rng(3)
    bs, P, ns = 4, 36, 7
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    imgs, masks, vis, flow, occ, feats = t(bs, 3, P, 1), t(bs, 1, P, 1), t(bs, 1, P, 1), t(bs, 2, P, 1), t(bs, 1, P, 1), t(bs, 16, P, 1)
    inds = torch.from_numpy(rng.integers(0, P, (bs, ns)))
    r = PL.obs_to_rays({}, inds, imgs, masks, vis, flow, occ, feats)
    for i in range(bs):           # moda.py:1244-1259
        assert torch.equal(r['img_at_samp'][i], imgs[i].view(3, -1).T[inds[i]])
        assert torch.equal(r['sil_at_samp'][i], masks[i].view(-1, 1)[inds[i]])
        assert torch.equal(r['flo_at_samp'][i], flow[i].view(2, -1).T[inds[i]])
        assert torch.equal(r['feats_at_samp'][i], feats[i].view(16, -1).T[inds[i]])
    # line form: R single pixels, each from the row batch_map names
    R = 9
    bm = torch.from_numpy(rng.integers(0, bs, (R,)))
    li = torch.from_numpy(rng.integers(0, P, (R, 1)))
    q = PL.obs_to_rays_line({}, li, imgs, masks, vis, flow, occ, feats, bm)
    for j in range(R):
        assert q['img_at_samp'].shape == (R, 1, 3)
        assert torch.equal(q['img_at_samp'][j, 0], imgs[bm[j], :, li[j, 0], 0])
        assert torch.equal(q['cfd_at_samp'][j, 0], occ[bm[j], :, li[j, 0], 0])
        assert torch.equal(q['feats_at_samp'][j, 0], feats[bm[j], :, li[j, 0], 0])


def test_line_dataset_reproduces_the_reference_reader(tmp_path):
    """G22: the files moda_amd writes were read by the REFERENCE's utils.io.LineDataset.__getitem__ (utils/io.py:380-454) in
    the build container; moda_amd's LineDataset, on the same files and the same numpy random stream, must return the same
    items -- every key, shape, dtype and value, including the default-camera fallback for a frame without a camera file."""
    from helpers import golden
    g = golden("g22_pixel_lines")
    n_frames, img_size, W, indices = 9, 6, 6, [0, 7, 13, 24, 47, 30]
    pix = str(tmp_path / "Pixels" / "Full-Resolution" / "syn")
    cam = str(tmp_path / "Cameras" / "Full-Resolution" / "syn")
    PL.write_synthetic_sequence(pix, 22, n_frames, img_size, W)
    PL.write_synthetic_cameras(cam, 22, n_frames, skip=(5,))
    ds = PL.LineDataset(pix, n_frames, img_size, rtklist=[os.path.join(cam, '%05d.txt' % i) for i in range(n_frames)], dataid=3)
    assert len(ds) == 48
    np.random.seed(22)                                   # the reference draws dframe from numpy's global stream (:424)
    keys = sorted({k.split("_", 1)[1] for k in g})
    assert set(keys) == set(PL.LINE_KEYS) | {"rtk", "kaug", "dataid", "frameid", "lineid"}
    fallback = 0
    for n, idx in enumerate(indices):
        e = ds[idx]
        assert set(e) == set(keys)
        for k in keys:
            want = g[f"item{n}_{k}"]
            got = np.asarray(e[k])
            assert got.shape == want.shape and got.dtype == want.dtype, (n, k, got.shape, want.shape, got.dtype, want.dtype)
            assert np.array_equal(got, want), (n, k)
        fallback += int(np.array_equal(np.asarray(e["rtk"])[0, 0], PL.default_camera()))
    assert 0 < fallback < len(indices)                   # both the camera files and the fallback were exercised
