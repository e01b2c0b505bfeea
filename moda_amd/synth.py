"""Deterministic synthetic scenes for the MoDA rendering path (numpy only, no RNG library state).

Shapes and distributions follow SURVEY.md section 8(d): rays in the object's
normalised frame (near 0.1 / far 0.5), per-frame pose / environment codes
replicated per ray exactly as moda.update_rays does (reference
nnutils/moda.py:1302-1310), bones in generate_bones layout
(geom_utils.py:841-855), per-frame unit dual quaternions built the way
DQ_RTHead builds them (nerf.py:263-276), and nn.Linear-law weights
U(-1/sqrt(fan_in), 1/sqrt(fan_in)).

Every value is a pure function of (seed, name, index) through a splitmix64
counter hash, so fixtures need to store only outputs and the GPU box can
regenerate the same inputs bit-for-bit.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _key(seed, name):
    return np.uint64((int(seed) * 0x9E3779B1 + zlib.crc32(name.encode())) & 0xFFFFFFFFFFFFFFFF)


def uniform(seed, name, shape):
    """float32 uniforms in [0,1) with 24 random bits."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        base = _splitmix64(np.full(1, _key(seed, name), dtype=np.uint64))[0]
        h = _splitmix64(np.arange(n, dtype=np.uint64) + base)
    return ((h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).reshape(shape)


def normal(seed, name, shape):
    """float32 standard normals (Box-Muller in float64, then rounded)."""
    u1 = uniform(seed, name + "/u1", shape).astype(np.float64)
    u2 = uniform(seed, name + "/u2", shape).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def linear_init(seed, name, out_f, in_f):
    bound = 1.0 / np.sqrt(in_f)
    w = (uniform(seed, name + ".weight", (out_f, in_f)) * 2 - 1) * np.float32(bound)
    b = (uniform(seed, name + ".bias", (out_f,)) * 2 - 1) * np.float32(bound)
    return w.astype(np.float32), b.astype(np.float32)


def nerf_params(seed, name, D=8, W=256, in_channels_xyz=63, in_channels_dir=27, out_channels=3,
                skips=(4,), init_beta=0.01):
    """State-dict-named parameters of a reference NeRF module (nerf.py:109-140)."""
    p = {}
    for i in range(D):
        fan = in_channels_xyz if i == 0 else (W + in_channels_xyz if i in skips else W)
        w, b = linear_init(seed, f"{name}.xyz_encoding_{i+1}.0", W, fan)
        p[f"xyz_encoding_{i+1}.0.weight"], p[f"xyz_encoding_{i+1}.0.bias"] = w, b
    p["xyz_encoding_final.weight"], p["xyz_encoding_final.bias"] = linear_init(seed, name + ".final", W, W)
    p["dir_encoding.0.weight"], p["dir_encoding.0.bias"] = linear_init(seed, name + ".dir", W // 2, W + in_channels_dir)
    p["sigma.weight"], p["sigma.bias"] = linear_init(seed, name + ".sigma", 1, W)
    p["rgb.0.weight"], p["rgb.0.bias"] = linear_init(seed, name + ".rgb", out_channels, W // 2)
    p["beta"] = np.asarray([init_beta], dtype=np.float32)
    return p


def _q_mul(a, b):
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack((aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw), -1)


def frame_dual_quats(seed, name, n_frames, B, rot=0.1, trans=0.02):
    """(F, B*8) unit dual quaternions [r, 0.5 (0,t) (x) r] -- DQ_RTHead's construction (nerf.py:263-276)."""
    r = np.asarray([1, 0, 0, 0], np.float32) + np.float32(rot) * normal(seed, name + "/r", (n_frames, B, 4))
    r = r / np.sqrt((r * r).sum(-1, keepdims=True))
    t = np.float32(trans) * normal(seed, name + "/t", (n_frames, B, 3))
    tq = np.concatenate([np.zeros_like(t[..., :1]), t], -1)
    d = np.float32(0.5) * _q_mul(tq, r)
    return np.concatenate([r, d], -1).reshape(n_frames, B * 8).astype(np.float32)


def make_rays(seed, N, B, rays_per_frame=256, with_app=False):
    """The `rays` dict of render_rays (rendering.py:57-60, 291, 301, 364-372), float32, ray-major."""
    n_frames = (N + rays_per_frame - 1) // rays_per_frame
    fid = np.arange(N) // rays_per_frame
    d = np.float32(0.1) * normal(seed, "rays_d", (N, 3)) + np.asarray([0, 0, 1], np.float32)
    d = d / np.sqrt((d * d).sum(-1, keepdims=True))
    o = np.float32(0.05) * normal(seed, "rays_o", (N, 3)) - np.asarray([0, 0, 0.3], np.float32)
    rays = {
        "rays_o": o.astype(np.float32),
        "rays_d": d.astype(np.float32),
        "near": np.full((N, 1), 0.1, np.float32),
        "far": np.full((N, 1), 0.5, np.float32),
        "xys": uniform(seed, "xys", (N, 2)) * np.float32(512),
        "time_embedded": normal(seed, "time_embedded", (n_frames, 128))[fid],
        "env_code": normal(seed, "env_code", (n_frames, 64))[fid],
    }
    if B > 0:
        rays["bone_rts"] = frame_dual_quats(seed, "bone_rts", n_frames, B)[fid]
    if with_app:
        rays["appearance_code"] = normal(seed, "appearance_code", (n_frames, 128))[fid]
    return {k: np.ascontiguousarray(v) for k, v in rays.items()}


def make_bones(seed, B):
    """generate_bones layout [center ; 1,0,0,0 ; 0,0,0] (geom_utils.py:841-855), centres ~N(0,0.1^2)."""
    bones = np.zeros((B, 10), np.float32)
    bones[:, :3] = np.float32(0.1) * normal(seed, "bones", (B, 3))
    bones[:, 3] = 1
    return bones


def make_models(seed, B=25, with_skin=True, with_feat=False, with_vis=False, with_app=False, beta=0.1,
                perturb_bones=False, with_dis=False):
    """Parameter sets of the `models` dict moda.__init__ builds (moda.py:271-348,444-449), as numpy dicts."""
    m = {"coarse": nerf_params(seed, "coarse", in_channels_dir=27 + 64 + (128 if with_app else 0), init_beta=beta)}
    if B > 0:
        bones = make_bones(seed, B)
        if perturb_bones:  # non-trivial orientations / scales, exercises vec_to_sim3 fully
            bones[:, 3:7] += np.float32(0.3) * normal(seed, "bones/q", (B, 4))
            bones[:, 7:10] = np.float32(0.3) * normal(seed, "bones/s", (B, 3))
        m["bones_rst"] = bones
        m["skin_aux"] = np.asarray([0, 10], np.float32)
        if with_skin:
            m["nerf_skin"] = nerf_params(seed, "nerf_skin", D=5, W=64, in_channels_xyz=63 + 128,
                                         in_channels_dir=0, out_channels=B)
            m["rest_pose_code"] = normal(seed, "rest_pose_code", (1, 128))
    if with_dis:      # residual displacement field (moda.py:334-340); a small last layer: displacements of a few cm
        m["nerf_dis"] = nerf_params(seed, "nerf_dis", D=5, W=128, in_channels_xyz=63 + 128, in_channels_dir=0, out_channels=3)
        m["nerf_dis"]["rgb.0.weight"] = m["nerf_dis"]["rgb.0.weight"] * np.float32(0.2)
        m["nerf_dis"]["rgb.0.bias"] = m["nerf_dis"]["rgb.0.bias"] * np.float32(0.2)
    if with_feat:
        m["nerf_feat"] = nerf_params(seed, "nerf_feat", D=5, W=128, in_channels_dir=0, out_channels=16, init_beta=1.0)
    if with_vis:
        m["nerf_vis"] = nerf_params(seed, "nerf_vis", D=5, W=64, in_channels_dir=0, out_channels=1)
    return m


def make_corresp_rays(seed, N, B, rays_per_frame=256, img_size=512):
    """Extra ray keys of the paired-frame correspondence / loss block of inference_deform (rendering.py:345-360, 439-571):
    target / dense-target bone poses and cameras (rtk_vec = [R 9 | T 3 | Kinv 9], moda.py:1281-1290) and the observed
    per-pixel signals."""
    n_frames = (N + rays_per_frame - 1) // rays_per_frame
    fid = np.arange(N) // rays_per_frame
    out = {}
    for tag in ("target", "dentrg"):
        out["bone_rts_" + tag] = frame_dual_quats(seed, "bone_rts_" + tag, n_frames, B)[fid]
        q = np.asarray([1, 0, 0, 0], np.float32) + np.float32(0.05) * normal(seed, "rtk/q/" + tag, (n_frames, 4))
        q = q / np.sqrt((q * q).sum(-1, keepdims=True))
        r, i, j, k = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = np.stack([1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r),
                      2 * (i * j + k * r), 1 - 2 * (i * i + k * k), 2 * (j * k - i * r),
                      2 * (i * k - j * r), 2 * (j * k + i * r), 1 - 2 * (i * i + j * j)], -1)
        T = np.asarray([0, 0, 1.0], np.float32) + np.float32(0.05) * normal(seed, "rtk/t/" + tag, (n_frames, 3))
        fx, px = np.float32(400.0), np.float32(img_size / 2)
        Kinv = np.tile(np.asarray([1 / fx, 0, -px / fx, 0, 1 / fx, -px / fx, 0, 0, 1], np.float32), (n_frames, 1))
        out["rtk_vec_" + tag] = np.concatenate([R, T, Kinv], -1).astype(np.float32)[fid]
    out["img_at_samp"] = uniform(seed, "img_at_samp", (N, 3))
    out["sil_at_samp"] = (uniform(seed, "sil_at_samp", (N, 1)) < 0.6).astype(np.float32)
    out["vis_at_samp"] = (uniform(seed, "vis_at_samp", (N, 1)) < 0.9).astype(np.float32)
    out["flo_at_samp"] = np.float32(0.1) * normal(seed, "flo_at_samp", (N, 2))
    cfd = uniform(seed, "cfd_at_samp", (N, 1))
    out["cfd_at_samp"] = np.where(cfd < 0.2, 0, cfd).astype(np.float32)
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


def make_feat_rays(seed, N, rays_per_frame=256, img_size=512, n_feat=16):
    """Ray keys of the CSE feature-matching / keypoint-reprojection heads (rendering.py:417-437, 573-578): the
    current frame's camera rtk_vec = [R 9 | T 3 | Kinv 9] (moda.py:1281-1290) and the observed pixel features."""
    n_frames = (N + rays_per_frame - 1) // rays_per_frame
    fid = np.arange(N) // rays_per_frame
    q = np.asarray([1, 0, 0, 0], np.float32) + np.float32(0.05) * normal(seed, "rtk/q/cur", (n_frames, 4))
    q = q / np.sqrt((q * q).sum(-1, keepdims=True))
    r, i, j, k = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r),
                  2 * (i * j + k * r), 1 - 2 * (i * i + k * k), 2 * (j * k - i * r),
                  2 * (i * k - j * r), 2 * (j * k + i * r), 1 - 2 * (i * i + j * j)], -1)
    T = np.asarray([0, 0, 1.0], np.float32) + np.float32(0.05) * normal(seed, "rtk/t/cur", (n_frames, 3))
    fx, px = np.float32(400.0), np.float32(img_size / 2)
    Kinv = np.tile(np.asarray([1 / fx, 0, -px / fx, 0, 1 / fx, -px / fx, 0, 0, 1], np.float32), (n_frames, 1))
    f = normal(seed, "feats_at_samp", (N, n_feat))
    f = f / np.sqrt((f * f).sum(-1, keepdims=True))
    out = {"rtk_vec": np.concatenate([R, T, Kinv], -1).astype(np.float32)[fid], "feats_at_samp": f.astype(np.float32)}
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


def make_cameras(seed, n_frames, img_size=512):
    """Per-frame cameras as moda.convert_root_pose / update_rays hand them to raycast (geom_utils.py:746-794):
    Rmat (F,3,3), Tmat (F,3), Kinv (F,3,3), near_far (F,2)."""
    q = np.asarray([1, 0, 0, 0], np.float32) + np.float32(0.2) * normal(seed, "cam/q", (n_frames, 4))
    q = q / np.sqrt((q * q).sum(-1, keepdims=True))
    r, i, j, k = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r),
                  2 * (i * j + k * r), 1 - 2 * (i * i + k * k), 2 * (j * k - i * r),
                  2 * (i * k - j * r), 2 * (j * k + i * r), 1 - 2 * (i * i + j * j)], -1).reshape(n_frames, 3, 3)
    T = np.asarray([0, 0, 1.0], np.float32) + np.float32(0.1) * normal(seed, "cam/t", (n_frames, 3))
    f = np.float32(400.0) + np.float32(20.0) * normal(seed, "cam/f", (n_frames,))
    px = np.float32(img_size / 2) + np.float32(5.0) * normal(seed, "cam/p", (n_frames, 2))
    Kinv = np.zeros((n_frames, 3, 3), np.float32)
    Kinv[:, 0, 0] = 1 / f; Kinv[:, 1, 1] = 1 / f; Kinv[:, 0, 2] = -px[:, 0] / f; Kinv[:, 1, 2] = -px[:, 1] / f; Kinv[:, 2, 2] = 1
    nf = np.stack([np.full(n_frames, 0.2, np.float32), np.full(n_frames, 1.8, np.float32)], -1) \
        + np.float32(0.05) * uniform(seed, "cam/nf", (n_frames, 2))
    return {"Rmat": R.astype(np.float32), "Tmat": T.astype(np.float32), "Kinv": Kinv, "near_far": nf.astype(np.float32)}


def evaluate_mlp_inputs(seed, N, S):
    """Inputs of the evaluate_mlp fixture (G18): sample positions, ray directions and the per-ray code rows."""
    d = normal(seed, "g18/dir", (N, 3))
    d = d / np.sqrt((d * d).sum(-1, keepdims=True))
    return dict(xyz=np.float32(0.3) * normal(seed, "g18/xyz", (N, S, 3)), dirs=d.astype(np.float32),
                env=normal(seed, "g18/env", (N, 64)), app=normal(seed, "g18/app", (N, 128)),
                tcode=normal(seed, "g18/tcode", (N, 128)), rest=normal(seed, "g18/rest", (1, 128)))


def make_unc_rays(seed, N, rays_per_frame):
    """ts / vid_code / xysn as moda.update_rays builds them for the uncertainty head (moda.py:1316-1327), per ray."""
    n_frames = (N + rays_per_frame - 1) // rays_per_frame
    fid = np.arange(N) // rays_per_frame
    ts = (np.float32(2.0) * uniform(seed, "g19/ts", (n_frames, 1)) - 1)[fid]
    vid = normal(seed, "g19/vid_code", (2, 32))[fid % 2]
    xysn = (uniform(seed, "g19/xysn", (N, 2)) - np.float32(0.5)) * np.float32(1.2)
    return {"ts": ts.astype(np.float32), "vid_code": vid.astype(np.float32), "xysn": xysn.astype(np.float32)}


def checkpoint_fill(key, shape, seed=20):
    """Deterministic value of tensor `key` of the checkpoint fixture (G20): the generator fills the reference model with
    these values, the tests rebuild the same params_*.pth from the fixture's key -> shape map."""
    if key.startswith("nerf_body_rts.0."):             # nn.Sequential(self.pose_code, head): the SAME module under two names
        key = "pose_code." + key[len("nerf_body_rts.0."):]
    if key == "alpha":
        return np.asarray([10.0], np.float32)
    if key == "skin_aux":
        return np.asarray([0.0, 10.0], np.float32)
    if key.endswith("beta"):
        return np.asarray([1.0 if ("feat" in key or "unc" in key) else 0.1], np.float32)
    if key == "near_far":
        a = np.zeros(shape, np.float32)
        a[:, 0], a[:, 1] = 0.1, 0.5
        return a
    if key == "bones":
        b = make_bones(seed, shape[0])
        b[:, 3:7] += np.float32(0.3) * normal(seed, "g20/bones/q", (shape[0], 4))
        b[:, 7:10] = np.float32(0.3) * normal(seed, "g20/bones/s", (shape[0], 3))
        return b
    fan = shape[-1] if len(shape) > 1 else 64
    scale = np.float32(1.0 / np.sqrt(fan))
    if key.startswith("nerf_body_rts.1.rgb"):          # small pose-head outputs: near-identity transforms
        scale = scale * np.float32(0.05)
    a = (uniform(seed, "g20/" + key, tuple(shape)) * 2 - 1) * scale
    if key == "nerf_body_rts.1.rgb.0.bias":
        a = a + np.tile(np.asarray([0, 0, 0, 1, 0, 0, 0], np.float32), shape[0] // 7)
    return a.astype(np.float32)
