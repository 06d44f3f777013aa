"""Fused multiply-add ``a * b + c`` with broadcasting-aware gradients.

Public surface mirrors the reference module torch_utils/ops/fma.py (``fma`` :15-16, gradient rules
:27-46, ``_unbroadcast`` :50-58). The shape pattern ``modulated_conv2d`` produces
(training/networks.py:77: activations [N,C,H,W] x per-sample channel scales [N,C,1,1] + noise
[N,1,H,W] or [H,W]) runs on ``pasta_scale_add`` / ``pasta_plane_dot`` (csrc/planes.hip); any other
broadcast pattern uses ``torch.addcmul`` on the same device.
"""

import torch

from . import _native

#----------------------------------------------------------------------------

def fma(a, b, c): # => a * b + c
    return _FusedMultiplyAdd.apply(a, b, c)

#----------------------------------------------------------------------------

def _plane_pattern(a, b, c):
    """True when (a, b, c) is the demodulate-and-add-noise pattern the HIP kernels cover."""
    if not (a.device.type == 'cuda' and a.dtype in _native.DTYPE_CODE and a.dtype != torch.float64 and a.ndim == 4 and a.is_contiguous()):
        return False
    n, ch, h, w = a.shape
    if b.dtype not in (a.dtype, torch.float32) or tuple(b.shape) != (n, ch, 1, 1):
        return False
    if c.dtype != a.dtype or tuple(c.shape) not in ((n, 1, h, w), (h, w), (1, 1, h, w)):
        return False
    return a.numel() > 0

def scale_planes(x, s, noise=None):
    """x[n,c,:,:] * s[n,c] (+ noise broadcast over channels) in one pass; NCHW fp32 / fp16 / bf16 on the GPU (the scales
    are fp32, the noise has x's type)."""
    _native.require_gpu(x, 'scale_add')
    n, ch, h, w = x.shape
    x = x.contiguous()
    s = s.reshape(n * ch).float().contiguous() if s is not None else None
    per_sample = 0
    if noise is not None:
        per_sample = int(noise.ndim == 4 and noise.shape[0] == n and n > 1)
        noise = noise.to(x.dtype).contiguous()
    y = torch.empty_like(x)
    row = None      # 131 000 four-instruction waves: a per-wave commit doubles the kernel (measured 1.34 -> 2.91 ms per step); the consumer scans
    with torch.cuda.device(x.device):
        st = _native.lib().pasta_scale_add(_native.ptr(x), _native.ptr(s), _native.ptr(noise), _native.ptr(y), _native.dtype_code(x, 'scale_add'),
                                           n, ch, h * w, per_sample, _native.stream(), _native.ptr(row))
    _native.check(st)
    return _native.amax_attach(y, row)

def plane_dot(p, q=None):
    """out[n,c] = sum_hw p*q (or sum_hw p) as fp32; NCHW fp32 / fp16 / bf16 on the GPU, fixed summation order."""
    _native.require_gpu(p, 'plane_dot')
    n, ch, h, w = p.shape
    p = p.contiguous()
    q = q.to(p.dtype).contiguous() if q is not None else None
    out = torch.empty([n, ch], dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        st = _native.lib().pasta_plane_dot(_native.ptr(p), _native.ptr(q), _native.ptr(out), _native.dtype_code(p, 'plane_dot'), n * ch, h * w,
                                           _native.stream())
    _native.check(st)
    return out

class _FusedMultiplyAdd(torch.autograd.Function): # a * b + c
    @staticmethod
    def forward(ctx, a, b, c):
        ctx.fast = _plane_pattern(a, b, c)
        if ctx.fast:
            out = scale_planes(a, b, c)
        else:
            out = torch.addcmul(c, a, b)
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da = db = dc = None
        fast = ctx.fast and dout.dtype == a.dtype and not torch.is_grad_enabled()
        if ctx.needs_input_grad[0]:
            da = scale_planes(dout, b) if fast else _unbroadcast(dout * b, a.shape)
        if ctx.needs_input_grad[1]:
            db = plane_dot(dout, a).reshape(b.shape).to(b.dtype) if fast else _unbroadcast(dout * a, b.shape)
        if ctx.needs_input_grad[2]:
            dc = _unbroadcast(dout, ctx.c_shape)
        return da, db, dc

#----------------------------------------------------------------------------

def _unbroadcast(x, shape):
    """Sum ``x`` down to ``shape`` (the adjoint of broadcasting ``shape`` up to ``x.shape``)."""
    extra_dims = x.ndim - len(shape)
    assert extra_dims >= 0
    dim = [i for i in range(x.ndim) if x.shape[i] > 1 and (i < extra_dims or shape[i - extra_dims] == 1)]
    if len(dim):
        x = x.sum(dim=dim, keepdim=True)
    if extra_dims:
        x = x.reshape(-1, *x.shape[extra_dims + 1:])
    assert x.shape == shape
    return x

#----------------------------------------------------------------------------
