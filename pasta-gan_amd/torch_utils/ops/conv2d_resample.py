"""2-D convolution with optional x``up`` / ÷``down`` resampling.

Public surface mirrors the reference module torch_utils/ops/conv2d_resample.py
(``conv2d_resample`` :59-154, ``_get_weight_shape`` :21-25, ``_conv2d_wrapper`` :29-54): the same
six-way decomposition into a dense convolution (``conv2d_gradfix`` -> fp32 MFMA implicit GEMM) and
an ``upfirdn2d`` low-pass, with padding applied once, up front. The cuDNN channels_last
work-around of the reference (:38-50) has no counterpart here.
"""

import torch

from . import conv2d_gradfix
from . import upfirdn2d
from .upfirdn2d import _parse_padding
from .upfirdn2d import _get_filter_size

#----------------------------------------------------------------------------

def _get_weight_shape(w):
    return [int(sz) for sz in w.shape]

def _conv2d_wrapper(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True, wgain=1.0, modulation=None):
    """Dispatch to ``conv2d`` / ``conv_transpose2d``. Both are correlations, so a true convolution
    (``flip_weight=False``) mirrors the taps first."""
    if not flip_weight:
        w = w.flip([2, 3])
    if modulation is not None:      # forward-only modulated convolution: (styles, dcoefs or None, per_sample)
        assert groups == 1 and wgain == 1.0
        return conv2d_gradfix.modulated_conv2d_forward(x, w, modulation[0], modulation[1], stride=stride, padding=padding, transposed=transpose,
                                                       per_sample=modulation[2])
    if transpose:
        return conv2d_gradfix.conv_transpose2d(x, w, stride=stride, padding=padding, groups=groups, wgain=wgain)
    return conv2d_gradfix.conv2d(x, w, stride=stride, padding=padding, groups=groups, wgain=wgain)

#----------------------------------------------------------------------------

def conv2d_resample_bias_act(x, w, b=None, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False,
                             act='linear', alpha=None, gain=None, clamp=None, wgain=1.0, residual=None, passthrough=False):
    """``bias_act(conv2d_resample(x, w, ...), b, act, alpha, gain, clamp)`` -- the body of ``Conv2dLayer.forward``
    (reference training/networks.py:170-179). When the dense convolution is the last step of the resampling
    decomposition (no upsampling), bias / activation / gain / clamp ride in its epilogue; otherwise the two ops run
    one after the other. ``wgain``: the convolution uses ``w * wgain`` (``Conv2dLayer``'s weight gain) without a
    multiplication kernel of its own.  ``residual`` (extension, shape of the output): added to the convolution before
    the bias -- also in the epilogue when the convolution is fused.  ``passthrough=True`` (extension) returns ``(y, x')``: ``x'`` is ``x``
    again, to be given to the other consumers of ``x`` so that their gradient joins this layer's input gradient in that launch's epilogue
    (``conv2d_gradfix._ConvBiasActHip``).  In a downsampling layer the FILTER is what reads ``x``: it hands ``x`` on and takes the gradient
    as the addend of its backward launch (``upfirdn2d._Upfirdn2dHip``); an upsampling layer returns ``x`` itself (nothing joined)."""
    from . import bias_act
    fusable = up == 1 and x.dtype in conv2d_gradfix.IO_CODES and x.device.type == 'cuda' and act in conv2d_gradfix.FUSABLE_ACTS
    if passthrough and (up != 1 or (down != 1 and not fusable)):
        return conv2d_resample_bias_act(x, w, b=b, f=f, up=up, down=down, padding=padding, groups=groups, flip_weight=flip_weight, flip_filter=flip_filter,
                                        act=act, alpha=alpha, gain=gain, clamp=clamp, wgain=wgain, residual=residual), x
    again = x
    if fusable:
        out_channels, in_channels_per_group, kh, kw = _get_weight_shape(w)
        fw, fh = _get_filter_size(f)
        px0, px1, py0, py1 = _parse_padding(padding)
        if down > 1:
            px0 += (fw - down + 1) // 2
            px1 += (fw - down) // 2
            py0 += (fh - down + 1) // 2
            py1 += (fh - down) // 2
        wc = w if flip_weight else w.flip([2, 3])
        # (downsampling layers with ``passthrough``: the FILTER is what reads x, so it hands x on and takes the other consumers' gradient)
        if kw == 1 and kh == 1 and down > 1:          # decimate, then the fused 1x1 convolution
            x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter, passthrough=passthrough)
            if passthrough:
                x, again = x
            y = conv2d_gradfix.conv2d_bias_act(x, wc, b, groups=groups, act=act, alpha=alpha, gain=gain, clamp=clamp, wgain=wgain, residual=residual)
            return (y, again) if passthrough else y
        if down == 2 and conv2d_gradfix.pieces_available(x, f, wc, (px0, px1, py0, py1), groups):
            # round 5: the low-pass writes the operand pieces of the default arithmetic instead of an fp32 tensor; the strided convolution and
            # its weight gradient copy them (conv2d_gradfix._BlurConvS2Hip)
            return conv2d_gradfix.blur_conv2d_s2_bias_act(x, f, wc, (px0, px1, py0, py1), flip_filter=flip_filter, bias=b, act=act, alpha=alpha, gain=gain,
                                                          clamp=clamp, wgain=wgain, residual=residual, passthrough=passthrough)
        if down > 1:                                  # low-pass, then the fused strided convolution
            x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0, px1, py0, py1], flip_filter=flip_filter, passthrough=passthrough)
            if passthrough:
                x, again = x
            y = conv2d_gradfix.conv2d_bias_act(x, wc, b, stride=down, groups=groups, act=act, alpha=alpha, gain=gain, clamp=clamp, wgain=wgain, residual=residual)
            return (y, again) if passthrough else y
        if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:
            return conv2d_gradfix.conv2d_bias_act(x, wc, b, padding=[py0, px0], groups=groups, act=act, alpha=alpha, gain=gain, clamp=clamp, wgain=wgain,
                                                  residual=residual, passthrough=passthrough)
    if passthrough:
        return conv2d_resample_bias_act(x, w, b=b, f=f, up=up, down=down, padding=padding, groups=groups, flip_weight=flip_weight, flip_filter=flip_filter,
                                        act=act, alpha=alpha, gain=gain, clamp=clamp, wgain=wgain, residual=residual), x
    x = conv2d_resample(x=x, w=w, f=f, up=up, down=down, padding=padding, groups=groups, flip_weight=flip_weight, flip_filter=flip_filter,
                        wgain=wgain)
    if residual is not None:
        x = x + residual
    return bias_act.bias_act(x, b, act=act, alpha=alpha, gain=gain, clamp=clamp)

#----------------------------------------------------------------------------

def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False, wgain=1.0, modulation=None):
    """Convolve ``x`` [N,C,H,W] with ``w`` [O,C//groups,kh,kw], upsampling by ``up`` before and/or
    downsampling by ``down`` after, low-pass filtered with ``f`` (from ``upfirdn2d.setup_filter``).

    ``padding`` is relative to the upsampled image (int, [x, y] or [x0, x1, y0, y1]).
    ``flip_weight=True`` is correlation (``torch.nn.functional.conv2d``), ``False`` convolution.
    ``modulation`` (extension, forward only, groups == 1) = (styles [N,I], dcoefs [N,O] or None, per_sample): the dense
    convolution of the decomposition is ``conv2d_gradfix.modulated_conv2d_forward`` -- styles in the kernel's staging
    (shared weight; dcoefs are then left to the caller) or styles and dcoefs in the weight packing (per-sample weights)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and (w.dtype == x.dtype or w.dtype == torch.float32)    # fp32 master weights may meet 16-bit activations
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1
    assert isinstance(down, int) and down >= 1
    assert isinstance(groups, int) and groups >= 1
    out_channels, in_channels_per_group, kh, kw = _get_weight_shape(w)
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)

    # The filter's own footprint counts as padding (reference :94-104).
    if up > 1:
        px0 += (fw + up - 1) // 2
        px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2
        py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2
        px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2
        py1 += (fh - down) // 2
    pad = [px0, px1, py0, py1]
    pointwise = (kw == 1 and kh == 1)

    # 1x1 kernel, downsampling: decimate first, then mix channels on the small image.
    if pointwise and down > 1 and up == 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, padding=pad, flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight, wgain=wgain, modulation=modulation)

    # 1x1 kernel, upsampling: mix channels on the small image, then interpolate.
    if pointwise and up > 1 and down == 1:
        x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight, wgain=wgain, modulation=modulation)
        return upfirdn2d.upfirdn2d(x=x, f=f, up=up, padding=pad, gain=up ** 2, flip_filter=flip_filter)

    # Downsampling only: low-pass at full resolution, strided convolution.
    if down > 1 and up == 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=pad, flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, stride=down, groups=groups, flip_weight=flip_weight, wgain=wgain, modulation=modulation)

    # Upsampling (optionally followed by downsampling): transposed strided convolution, then low-pass.
    if up > 1:
        if groups == 1:
            w = w.transpose(0, 1)
        else:
            w = w.reshape(groups, out_channels // groups, in_channels_per_group, kh, kw)
            w = w.transpose(1, 2)
            w = w.reshape(groups * in_channels_per_group, out_channels // groups, kh, kw)
        px0 -= kw - 1
        px1 -= kw - up
        py0 -= kh - 1
        py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = _conv2d_wrapper(x=x, w=w, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, flip_weight=(not flip_weight), wgain=wgain, modulation=modulation)
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
        return x

    # No resampling and symmetric non-negative padding: a plain convolution.
    if up == 1 and down == 1 and px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:
        return _conv2d_wrapper(x=x, w=w, padding=[py0, px0], groups=groups, flip_weight=flip_weight, wgain=wgain, modulation=modulation)

    # Anything else: pad/upsample, convolve, downsample as three separate steps.
    x = upfirdn2d.upfirdn2d(x=x, f=(f if up > 1 else None), up=up, padding=pad, gain=up ** 2, flip_filter=flip_filter)
    x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight, wgain=wgain, modulation=modulation)
    if down > 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
    return x

#----------------------------------------------------------------------------
