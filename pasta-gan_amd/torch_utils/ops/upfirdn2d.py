"""2-D resampling (pad / upsample / FIR / downsample) on the MI355X HIP kernel.

Public surface mirrors the reference module torch_utils/ops/upfirdn2d.py
(``setup_filter`` :72, ``upfirdn2d`` :120, ``filter2d`` :272, ``upsample2d`` :308,
``downsample2d`` :347, helpers ``_parse_scaling`` :37, ``_parse_padding`` :46,
``_get_filter_size`` :57) so that callers -- ``training.networks``, ``training.augment``
and source embedded in pickles -- bind to it unchanged.  The arithmetic runs in
``pasta_upfirdn2d`` (csrc/upfirdn2d.hip) on the caller's current stream.
"""

import numpy as np
import torch

from . import _native

#----------------------------------------------------------------------------
# Argument normalisation.

def _parse_scaling(scaling):
    if isinstance(scaling, int):
        scaling = [scaling, scaling]
    assert isinstance(scaling, (list, tuple)) and len(scaling) == 2
    sx, sy = scaling
    assert isinstance(sx, int) and isinstance(sy, int) and sx >= 1 and sy >= 1
    return sx, sy

def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple))
    assert all(isinstance(v, int) for v in padding)
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    padx0, padx1, pady0, pady1 = padding
    return padx0, padx1, pady0, pady1

def _get_filter_size(f):
    """(width, height) of a filter tensor; ``None`` is the 1x1 identity."""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    fw, fh = int(f.shape[-1]), int(f.shape[0])
    assert fw >= 1 and fh >= 1
    return fw, fh

#----------------------------------------------------------------------------

def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Prepare a FIR filter for :func:`upfirdn2d` (reference: upfirdn2d.py:72-116).

    ``f`` may be ``None`` (identity), a scalar, a 1-D tap list or a 2-D kernel. A 1-D
    list with fewer than 8 taps is expanded to its outer product unless ``separable``
    says otherwise. Returns a float32 tensor on ``device``."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f.reshape(1)
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)

#----------------------------------------------------------------------------
# Native call + autograd.

# Optional measurement hook (bench.py): when set, called as hook(algorithmic_bytes, shape_key, launch) around every native
# launch; ``launch()`` performs it. None = no overhead.
launch_hook = None

def _launch(x, f2d, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain, add=None):
    """One ``pasta_upfirdn2d`` launch. ``f2d`` is a dense float32 [fh, fw] tensor on x's device.  ``add``: a tensor of the output's shape
    added to the result on its way out."""
    _native.require_gpu(x, 'upfirdn2d')
    if x.ndim != 4:
        raise RuntimeError('upfirdn2d: x must be rank 4')
    if f2d.dtype != torch.float32:
        raise RuntimeError('upfirdn2d: f must be float32')
    if f2d.device != x.device:
        raise RuntimeError('upfirdn2d: f must reside on the same device as x')
    n, c, ih, iw = x.shape
    fh, fw = f2d.shape
    ow = (iw * upx + padx0 + padx1 - fw + downx) // downx
    oh = (ih * upy + pady0 + pady1 - fh + downy) // downy
    if ow < 1 or oh < 1:
        raise RuntimeError('upfirdn2d: output must be at least 1x1')
    channels_last = x.stride(1) == 1 and c > 1 and x.is_contiguous(memory_format=torch.channels_last)
    if not (channels_last or x.is_contiguous()):
        x = x.contiguous()
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device,
                    memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    if y.numel() == 0:
        return y
    if add is not None:
        if tuple(add.shape) != tuple(y.shape):
            raise RuntimeError(f'upfirdn2d: the addend {tuple(add.shape)} does not have the output shape {tuple(y.shape)}')
        add = add.to(y.dtype)
        if add.stride() != y.stride():
            add = add.contiguous(memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    f2d = f2d.contiguous()
    row = _native.amax_slot(y) if y.numel() >= 1 << 16 else None
    def launch():
        with torch.cuda.device(x.device):
            st = _native.lib().pasta_upfirdn2d(
                _native.ptr(x), _native.ptr(f2d), _native.ptr(y), _native.dtype_code(x, 'upfirdn2d'),
                _native.i32x(*x.shape), _native.i64x(*x.stride()), _native.i32x(fh, fw),
                _native.i32x(*y.shape), _native.i64x(*y.stride()),
                upx, upy, downx, downy, padx0, padx1, pady0, pady1, int(bool(flip)), float(gain), _native.stream(), _native.ptr(row), _native.ptr(add))
        _native.check(st)
    if launch_hook is None:
        launch()
    else:       # algorithmic bytes of the launch: (numel_in + numel_out) * sizeof(T)
        launch_hook((x.numel() + y.numel()) * x.element_size(), (tuple(x.shape), upx, downx, fw), launch)
    return _native.amax_attach(y, row)

class _Upfirdn2dHip(torch.autograd.Function):
    """y = upfirdn2d(x, f); the gradient is the same operator with up/down exchanged and the
    filter mirrored (reference: upfirdn2d.py:246-264), so gradients of any order come for free."""

    @staticmethod
    def forward(ctx, x, f, cfg, add=None, passthrough=False):
        """``add`` (shape of the output): ``upfirdn2d(x, f) + add`` in the one launch.  ``passthrough=True`` returns ``(y, x)``: the second
        output is ``x`` again, for its OTHER consumers, whose gradient then arrives in the backward as ``dxp`` and is the ``add`` operand of
        the backward launch -- instead of an addition pass of autograd over two tensors (conv2d_gradfix._ConvBiasActHip has the same)."""
        upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain = cfg
        if f is None:
            f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
        assert f.ndim in [1, 2]
        if f.ndim == 2:
            y = _launch(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain, add=add)
        else:   # separable: a row pass then a column pass, sqrt(gain) each
            g = float(np.sqrt(gain))
            y = _launch(x, f.unsqueeze(0), upx, 1, downx, 1, padx0, padx1, 0, 0, flip, g)
            y = _launch(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, pady0, pady1, flip, g, add=add)
        ctx.save_for_backward(f)
        ctx.cfg = cfg
        ctx.in_hw = (x.shape[2], x.shape[3])
        if passthrough:
            ctx.set_materialize_grads(False)
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, dxp=None):
        f, = ctx.saved_tensors
        upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain = ctx.cfg
        ih, iw = ctx.in_hw
        if dy is None:                  # only the pass-through output was differentiated
            return dxp, None, None, None, None
        oh, ow = dy.shape[2], dy.shape[3]
        fw, fh = _get_filter_size(f)
        dx = None
        if ctx.needs_input_grad[0]:
            gcfg = (downx, downy, upx, upy,
                    fw - padx0 - 1, iw * upx - ow * downx + padx0 - upx + 1,
                    fh - pady0 - 1, ih * upy - oh * downy + pady0 - upy + 1,
                    not flip, gain)
            dx = _Upfirdn2dHip.apply(dy, f, gcfg, dxp) if dxp is not None else _Upfirdn2dHip.apply(dy, f, gcfg)
        assert not ctx.needs_input_grad[1]
        dadd = dy if len(ctx.needs_input_grad) > 3 and ctx.needs_input_grad[3] else None
        return dx, None, None, dadd, None

#----------------------------------------------------------------------------

def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda', passthrough=False):
    """Pad, upsample, filter and downsample a batch of 2-D images (reference: upfirdn2d.py:120-164).

    ``x``: [N, C, H, W] float32/float16/float64 on the GPU. ``f``: float32 [fh, fw], separable
    [taps], or ``None``. ``padding`` is relative to the upsampled image; negative values crop.
    ``flip_filter=False`` is true convolution. ``impl`` is kept for call compatibility: ``'cuda'``
    is the HIP kernel (PyTorch-ROCm names the device 'cuda'); ``'ref'`` is not provided here."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'ref':
        raise NotImplementedError("upfirdn2d(impl='ref'): this package has no PyTorch-op fallback; "
                                  "the CPU restatement used for testing is oracle/ref_ops.py")
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2])
    assert f is None or not f.requires_grad
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    cfg = (upx, upy, downx, downy, padx0, padx1, pady0, pady1, bool(flip_filter), gain)
    if passthrough:             # (y, x'): own extension, see _Upfirdn2dHip.forward
        if torch.is_grad_enabled() and x.requires_grad:
            y, again = _Upfirdn2dHip.apply(x, f, cfg, None, True)
            hit = getattr(x, '_pasta_amax', None)
            if hit is not None and not again.is_inference():
                again._pasta_amax = hit
            return y, again
        return _Upfirdn2dHip.apply(x, f, cfg), x
    return _Upfirdn2dHip.apply(x, f, cfg)

#----------------------------------------------------------------------------
# Convenience wrappers: same output-size conventions as the reference (:272-382).

def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """FIR-filter keeping the input size (plus ``padding``)."""
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + fw // 2, padx1 + (fw - 1) // 2, pady0 + fh // 2, pady1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)

def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Upsample by ``up``; output size is a multiple of the input size (plus ``padding``)."""
    upx, upy = _parse_scaling(up)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw + upx - 1) // 2, padx1 + (fw - upx) // 2,
         pady0 + (fh + upy - 1) // 2, pady1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)

def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Downsample by ``down``; output size is a fraction of the input size (plus ``padding``)."""
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw - downx + 1) // 2, padx1 + (fw - downx) // 2,
         pady0 + (fh - downy + 1) // 2, pady1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)

#----------------------------------------------------------------------------
