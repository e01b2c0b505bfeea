"""fp16 mode: the overflow report of the fused kernels (include/moda_hip.h `moda_mlp_desc.overflow`).

The fp16-operand kernels never saturate silently: a launch in which a hidden activation (or, at packing time, a weight) does
not fit fp16 stores 1 into a flag word the caller supplies.  This module owns that word -- ONE pinned host int32 per process,
which the GPU writes through its device-visible address and the host reads WITHOUT synchronising -- and turns it into an
exception:

  * every fp16-mode entry of the package (`NeRF.fused`, `fused_warp`, `render_rays`) looks at the word first, so an overflow
    in an earlier, already executed call is raised at the next call at the latest;
  * `check()` synchronises the device and raises for everything enqueued so far: call it where a result is consumed.
"""
import torch

_FLAG = None


class Fp16Overflow(FloatingPointError):
    pass


def _flag():
    global _FLAG
    if _FLAG is None:
        _FLAG = torch.zeros(1, dtype=torch.int32).pin_memory()
    return _FLAG


def ptr():
    """Address of the flag word, for `moda_mlp_desc.overflow` / `moda_mlp_pack(..., overflow)`."""
    return _flag().data_ptr()


def tripped():
    """Has any fp16 launch that has EXECUTED so far reported an overflow?  A host read, no synchronisation."""
    return _FLAG is not None and int(_FLAG[0]) != 0


def reset():
    if _FLAG is not None:
        _FLAG[0] = 0


def poll():
    """Raise for an overflow reported by work that has already executed (cheap: one host load)."""
    if tripped():
        reset()
        raise Fp16Overflow("a fused fp16 launch produced a weight or hidden activation outside fp16's range (|v| >= 65520 or "
                           "NaN): its results are not valid -- use set_precision('bf16x3') or 'fp32' for this model")


def check():
    """Synchronise, then raise if any fp16 launch enqueued so far overflowed."""
    if _FLAG is not None:
        torch.cuda.synchronize()
    poll()
