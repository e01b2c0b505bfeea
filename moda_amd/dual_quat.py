"""The seven functions of the reference's nnutils/dual_quat.py with identical signatures, on the HIP library.

Quaternions are real-first; `dq` = [real quaternion (4), dual quaternion (4)].
"""
import torch

from . import _lib as L

_QMUL, _DQMUL, _NORMALIZE, _QCONJ, _CCONJ, _INVERSE, _QNORMALIZE = range(7)


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in ts)


def _autograd_op(op, a, b):
    """Per-frame callers (correct_rest_pose, correct_bones: geom_utils.py:933-972) differentiate through these
    few-element formulas; under autograd they are spelled with elementwise tensor ops (O(frames x bones) work)."""
    def qm(x, y):
        xw, xx, xy, xz = x.unbind(-1)
        yw, yx, yy, yz = y.unbind(-1)
        return torch.stack((xw * yw - xx * yx - xy * yy - xz * yz, xw * yx + xx * yw + xy * yz - xz * yy,
                            xw * yy - xx * yz + xy * yw + xz * yx, xw * yz + xx * yy - xy * yx + xz * yw), -1)
    if op == _QMUL:
        return qm(a, b)
    if op == _DQMUL:
        return torch.cat([qm(a[..., :4], b[..., :4]), qm(a[..., :4], b[..., 4:]) + qm(a[..., 4:], b[..., :4])], -1)
    if op == _NORMALIZE:
        return a / a[..., :4].norm(dim=-1, keepdim=True)
    if op == _QNORMALIZE:
        return a / a.norm(dim=-1, keepdim=True)
    if op == _QCONJ:
        return a * a.new_tensor([1, -1, -1, -1, 1, -1, -1, -1])
    if op == _CCONJ:
        return a * a.new_tensor([1, -1, -1, -1, -1, 1, 1, 1])
    return a * a.new_tensor([1, -1, -1, -1, 1, -1, -1, -1]) / (a[..., :4] ** 2).sum(-1, keepdim=True)


def _op(op, a, b, width, check_norm=False):
    assert a.shape[-1] == width
    if _needs_grad(a, b):
        return _autograd_op(op, L.dev(a), None if b is None else L.dev(b))
    shape = a.shape
    a2 = L.dev(a).reshape(-1, width)
    b2 = None
    if b is not None:
        assert b.shape[-1] == width
        b2 = L.dev(b).reshape(-1, width)
    out = torch.empty_like(a2)
    flag = torch.zeros(1, dtype=torch.int32, device=a2.device) if check_norm else None
    L.call("moda_dq_op", op, L.ptr(a2), L.ptr(b2), a2.shape[0], L.ptr(out), L.ptr(flag), L.stream())
    if check_norm:
        # the reference asserts on a singular quaternion (dual_quat.py:11,61); this is its host sync
        assert int(flag.item()) == 0, "singular (zero-norm) quaternion"
    return out.view(shape)


def q_normalize(q):
    """dual_quat.py:4-12"""
    return _op(_QNORMALIZE, q, None, 4, check_norm=True)


def q_mul(q1, q2):
    """dual_quat.py:14-31: Hamilton product q1 (x) q2"""
    return _op(_QMUL, q1, q2, 4)


def dq_mul(dq1, dq2):
    """dual_quat.py:33-49"""
    return _op(_DQMUL, dq1, dq2, 8)


def dq_normalize(dq):
    """dual_quat.py:51-62"""
    return _op(_NORMALIZE, dq, None, 8, check_norm=True)


def dq_quaternion_conjugate(dq):
    """dual_quat.py:65-74"""
    return _op(_QCONJ, dq, None, 8)


def dq_combined_conjugate(dq):
    """dual_quat.py:76-85"""
    return _op(_CCONJ, dq, None, 8)


def dq_inverse(dq):
    """dual_quat.py:87-94"""
    return _op(_INVERSE, dq, None, 8)
