"""Thin helpers shared by the op front-ends: library handle, dtype codes, pointers."""

import ctypes
import torch

from .. import custom_ops

DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.float64: 2, torch.bfloat16: 3}

_lib = None

def lib():
    """The loaded ``libpasta_hip.so``; raises if it cannot be built or loaded."""
    global _lib
    if _lib is None:
        _lib = custom_ops.get_plugin('pasta_hip')
    return _lib

def check(status):
    custom_ops.check(lib(), status)

def require_gpu(t, what):
    if t.device.type != 'cuda':
        raise RuntimeError(f'{what}: tensor is on {t.device}; the HIP path needs a GPU tensor '
                           f'(there is no CPU fallback in this package; the CPU restatement lives in oracle/)')

def ptr(t):
    """Device pointer of a tensor, or NULL for None."""
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)

def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

def dtype_code(t, what):
    try:
        return DTYPE_CODE[t.dtype]
    except KeyError:
        raise RuntimeError(f'{what}: unsupported dtype {t.dtype}') from None

def i32x(*vals):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])

def i64x(*vals):
    return (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])

# ---- producer-side maxima (include/pasta_hip.h, "producer-side maxima"): a kernel that writes an fp32 activation tensor leaves the
# tensor's largest magnitude in a zeroed 256-float row, and the row travels with the Python tensor object (conv2d_gradfix.tensor_amax
# finds it there), so the convolution that consumes the tensor under PASTA_MATH_F16X3 scans nothing.
_amax_pools = {}
import os as _os
_AMAX_PRODUCERS = _os.environ.get('PASTA_AMAX_PRODUCERS', '1') != '0'       # A/B switch: 0 = every consumer scans its operand
_AMAX_ROWS = 1024

def amax_slot(like):
    """A zeroed [256] fp32 row on ``like``'s device, or None when the running arithmetic has no use for it."""
    from . import conv2d_gradfix
    if not _AMAX_PRODUCERS:
        return None
    if conv2d_gradfix.conv_math not in ('default', 'f16x3') or like.dtype != torch.float32 or like.device.type != 'cuda':
        return None
    if torch.cuda.is_current_stream_capturing():
        # under hipGraph capture the row must be zeroed by a node of the graph itself: a replay reuses the same memory, and maxima only grow
        return torch.zeros([256], dtype=torch.float32, device=like.device)
    pool = _amax_pools.get(like.device)
    if pool is None or pool[1] >= _AMAX_ROWS:
        pool = [torch.zeros([_AMAX_ROWS, 256], dtype=torch.float32, device=like.device), 0]      # one fill per 1024 tensors
        _amax_pools[like.device] = pool
    row = pool[0][pool[1]]
    pool[1] += 1
    return row

def amax_attach(t, row):
    """Hang the producer's row on the tensor it describes (valid for the tensor's current version).  Inference tensors
    (``torch.inference_mode()``) track no version, so nothing is attached and their consumers scan."""
    if row is not None and not t.is_inference():
        try:
            t._pasta_amax = (t._version, t.data_ptr(), row, 'producer')      # a fourth field marks a producer's row (conv2d_gradfix.tensor_amax)
        except AttributeError:
            pass
    return t
