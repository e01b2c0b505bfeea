"""Import the MoDA reference hot path from /root/reference (THIS container only).

Test infrastructure for generating golden vectors; never imported by the
product package, bench.py's timed legs, or anything that runs on the GPU box.

The reference pulls in cv2 / trimesh / png / torchvision / pytorch3d at import
time (geom_utils.py:5-17, nerf.py:3-11, loss_utils.py:3-12).  None of them is
installed here.  cv2 / png / torchvision / trimesh are only needed as names, so
they are empty modules.  pytorch3d.transforms is an absent, un-vendored,
version-unpinned third-party dependency (.gitmodules:7-9, misc/moda.yml:4); the
four quaternion helpers reached on the neudbs path are restated here from their
published closed forms (real-first Hamilton product, conjugate, standard
rotation matrix scaled by 2/|q|^2, product with the real part made >= 0).
"""
import sys
import types

import torch

REF = "/root/reference"


def _quaternion_raw_multiply(a, b):
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    ow = aw * bw - ax * bx - ay * by - az * bz
    ox = aw * bx + ax * bw + ay * bz - az * by
    oy = aw * by - ax * bz + ay * bw + az * bx
    oz = aw * bz + ax * by - ay * bx + az * bw
    return torch.stack((ow, ox, oy, oz), -1)


def _standardize_quaternion(q):
    return torch.where(q[..., 0:1] < 0, -q, q)


def _quaternion_multiply(a, b):
    return _standardize_quaternion(_quaternion_raw_multiply(a, b))


def _quaternion_invert(q):
    return q * q.new_tensor([1, -1, -1, -1])


def _quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k),
            two_s * (i * j - k * r),
            two_s * (i * k + j * r),
            two_s * (i * j + k * r),
            1 - two_s * (i * i + k * k),
            two_s * (j * k - i * r),
            two_s * (i * k - j * r),
            two_s * (j * k + i * r),
            1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


def install_stubs():
    for name in ("cv2", "png", "torchvision"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if "trimesh" not in sys.modules:
        tm = types.ModuleType("trimesh")
        tm.Trimesh = type("Trimesh", (), {})
        sys.modules["trimesh"] = tm
    if "pytorch3d" not in sys.modules:
        p3d = types.ModuleType("pytorch3d")
        tr = types.ModuleType("pytorch3d.transforms")
        tr.quaternion_raw_multiply = _quaternion_raw_multiply
        tr.quaternion_multiply = _quaternion_multiply
        tr.quaternion_invert = _quaternion_invert
        tr.quaternion_to_matrix = _quaternion_to_matrix
        p3d.transforms = tr
        sys.modules["pytorch3d"] = p3d
        sys.modules["pytorch3d.transforms"] = tr


def import_reference():
    """Returns (rendering, nerf, geom_utils, dual_quat) modules of the reference."""
    install_stubs()
    for p in (REF + "/nnutils", REF + "/third_party", REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    import importlib

    rendering = importlib.import_module("nnutils.rendering")
    nerf = importlib.import_module("nnutils.nerf")
    geom = importlib.import_module("nnutils.geom_utils")
    dq = importlib.import_module("dual_quat")
    return rendering, nerf, geom, dq
