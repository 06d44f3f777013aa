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
