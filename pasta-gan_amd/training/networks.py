"""PASTA-GAN generator / discriminator layers on the MI355X HIP op layer.

Host-side mirror of the classes of the reference's ``training/networks.py`` that the shipped
training and test entry points construct: same class names, constructor
keywords, ``forward`` signatures and parameter / buffer names, so ``construct_class_by_name``
strings and state dicts interchange. Every tensor op with real traffic goes through
``torch_utils.ops`` (upfirdn2d, bias_act, conv2d_resample, fma) or the fused plane kernels
(``scale_planes``, ``spade_modulate``); small dense algebra (affine layers, demodulation
coefficients, minibatch statistics) stays on PyTorch-ROCm's BLAS.

Reference line numbers are cited per class as ``networks.py:<lines>``.
"""

import ctypes

import numpy as np
import torch
import torch.nn as nn

from torch_utils import misc
from torch_utils import persistence
from torch_utils.ops import conv2d_gradfix
from torch_utils.ops import conv2d_resample
from torch_utils.ops import upfirdn2d
from torch_utils.ops import bias_act
from torch_utils.ops import fma
from torch_utils.ops import _native

#----------------------------------------------------------------------------
# Plane-wise fused ops with autograd.

class _ScalePlanes(torch.autograd.Function):
    """y[n,c] = x[n,c] * s[n,c]: the activation-side modulation of networks.py:74."""
    @staticmethod
    def forward(ctx, x, s):
        ctx.save_for_backward(x, s)
        return fma.scale_planes(x, s)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        dx = ds = None
        if ctx.needs_input_grad[0]:
            dx = fma.scale_planes(dy, s)
        if ctx.needs_input_grad[1]:
            ds = fma.plane_dot(dy, x).reshape(s.shape).to(s.dtype)
        return dx, ds

_HIP_DTYPES = (torch.float32, torch.float16, torch.bfloat16)      # activation storage types of the fused plane kernels

def scale_planes(x, s):
    """x * s.reshape(N, C, 1, 1) in one pass (NCHW fp32 / fp16 / bf16 on the GPU, fp32 scales); otherwise a broadcast multiply."""
    if x.dtype in _HIP_DTYPES and x.device.type == 'cuda' and x.ndim == 4 and x.numel() > 0:
        return _ScalePlanes.apply(x, s.to(torch.float32).reshape(x.shape[0], x.shape[1]))
    return x * s.to(x.dtype).reshape(x.shape[0], -1, 1, 1)

class _SpadeModulate(torch.autograd.Function):
    """InstanceNorm(x) * (1 + gamma) + beta with statistics, normalisation and modulation in one
    kernel (networks.py:4371-4379: InstanceNorm2d(affine=False), eps 1e-5, biased variance).
    ``beta is None``: ``gamma`` is a [N, 2C, H, W] tensor holding gamma in its first C channels and beta in its last C (the
    output of ONE convolution with the concatenated conv_gamma / conv_beta weights); its gradient comes back as one tensor,
    so the two input gradients of that convolution accumulate inside its K loop instead of in an addition pass."""
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, post, shared=None, passthrough=False):
        """``passthrough=True`` returns ``(out, x)``: the second output is ``x`` again, for the block's OTHER normalisation of the same tensor,
        whose input gradient then arrives here as ``dxp`` and is added to ``dx`` by the backward kernel on its way out (``dx_add``) instead of
        by an addition pass of autograd over both."""
        n, c, h, w = x.shape
        x_in = x
        x = x.contiguous()
        fused = beta is None
        if fused and gamma.shape == (n, 2 * c, h, w) and gamma.stride()[1:] == (h * w, w, 1) and gamma.stride(0) >= 2 * c * h * w:
            pass            # gamma | beta as a channel slice of a wider tensor (the block's batched convolution): read at its sample stride
        else:
            gamma, shared = gamma.contiguous(), None
        if fused:
            assert gamma.shape == (n, 2 * c, h, w)
            beta_ptr, gstride = gamma.data_ptr() + x.element_size() * c * h * w, gamma.stride(0)
        else:
            beta = beta.contiguous()
            beta_ptr, gstride = beta.data_ptr(), 0
        out = torch.empty_like(x)
        stats = torch.empty([n * c, 2], dtype=torch.float32, device=x.device)
        act, gain, clamp = post                     # (2, gain, clamp): relu * gain with clamp on the way out; (0, 1, -1): none
        row = _native.amax_slot(out)
        with torch.cuda.device(x.device):
            st = _native.lib().pasta_spade_norm(_native.ptr(x), _native.ptr(gamma), beta_ptr, _native.ptr(out),
                                                _native.ptr(stats), _native.dtype_code(x, 'spade_norm'), n * c, h * w, float(eps), act, float(gain), float(clamp),
                                                c, gstride, _native.stream(), _native.ptr(row))
        _native.check(st)
        _native.amax_attach(out, row)
        ctx.save_for_backward(x, gamma, stats, beta if (act == 2 and not fused) else None)
        ctx.post, ctx.fused, ctx.shared = post, fused, shared
        if passthrough:
            ctx.set_materialize_grads(False)
            return out, x_in
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout, dxp=None):
        x, gamma, stats, beta = ctx.saved_tensors
        act, gain, clamp = ctx.post
        n, c, h, w = x.shape
        if dout is None:                            # only the pass-through output was differentiated
            return dxp, None, None, None, None, None, None
        dout = dout.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        if dxp is not None:
            dxp = dxp.to(x.dtype).contiguous() if dx is not None else None
        row = _native.amax_slot(dx) if dx is not None else None
        lib = _native.lib()
        if ctx.fused:       # gamma | beta and their gradients as channel halves of one tensor each
            half = x.element_size() * c * h * w
            dgb = grow = None
            if ctx.needs_input_grad[1]:
                if ctx.shared is not None:      # this block's slice of the gradient of the batched gamma | beta convolution (_SharedGrad)
                    holder, idx = ctx.shared
                    dgb, grow = holder.view(idx), holder.row
                else:
                    dgb = torch.empty([n, 2 * c, h, w], dtype=x.dtype, device=x.device)
                    grow = _native.amax_slot(dgb)                           # dgamma | dbeta is the dy of ONE convolution's backward
            if dx is not None or dgb is not None:
                with torch.cuda.device(x.device):
                    st = lib.pasta_spade_norm_bwd(_native.ptr(dout), _native.ptr(x), _native.ptr(gamma), _native.ptr(stats), _native.ptr(dx),
                                                  _native.ptr(dgb), dgb.data_ptr() + half if dgb is not None else None,
                                                  _native.dtype_code(x, 'spade_norm_bwd'), n * c, h * w,
                                                  gamma.data_ptr() + half, act, float(gain), float(clamp), c, gamma.stride(0),
                                                  dgb.stride(0) if dgb is not None else 2 * c * h * w,
                                                  _native.stream(), _native.ptr(row), _native.ptr(grow), _native.ptr(dxp))
                _native.check(st)
                if dx is not None:
                    _native.amax_attach(dx, row)
                if dgb is not None and ctx.shared is None:
                    _native.amax_attach(dgb, grow)
            elif dxp is not None:
                dx = dxp
            return dx, dgb, None, None, None, None, None
        dgamma = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        # without a fused activation d/dbeta is dout itself; with one it is dout through the activation, written by the kernel
        dbeta = None
        if act == 2 and (ctx.needs_input_grad[2] or dgamma is not None or dx is not None):
            dbeta = torch.empty_like(x)
        elif ctx.needs_input_grad[2]:
            dbeta = dout
        if dx is not None or dgamma is not None or act == 2:
            with torch.cuda.device(x.device):
                st = lib.pasta_spade_norm_bwd(_native.ptr(dout), _native.ptr(x), _native.ptr(gamma), _native.ptr(stats),
                                              _native.ptr(dx), _native.ptr(dgamma), _native.ptr(dbeta if act == 2 else None),
                                              _native.dtype_code(x, 'spade_norm_bwd'), n * c, h * w, _native.ptr(beta), act, float(gain), float(clamp), c, 0, 0, _native.stream(),
                                              _native.ptr(row), None, _native.ptr(dxp))
            _native.check(st)
            if dx is not None:
                _native.amax_attach(dx, row)
        return dx, dgamma, (dbeta if ctx.needs_input_grad[2] else None), None, None, None, None

class _SharedGrad:
    """ONE gradient tensor written in channel slices by several backward nodes: the three SPADE normalisations of a residual block leave
    dgamma | dbeta of their group in channels [g 2C, (g + 1) 2C) of the dy of the block's batched gamma | beta convolution, so that convolution's
    backward is one input-gradient and one weight-gradient launch and nothing is concatenated or added.  One row of producer maxima serves
    the three writers (a maximum does not care about the order)."""
    def __init__(self, like, groups):
        self.shape, self.dtype, self.device, self.groups = tuple(like.shape), like.dtype, like.device, groups
        self.buf = self.row = None

    def view(self, g):
        if self.buf is None:
            self.buf = torch.empty(self.shape, dtype=self.dtype, device=self.device)
            self.row = _native.amax_slot(self.buf)
        c = self.shape[1] // self.groups
        return self.buf.narrow(1, g * c, c)

class _SplitGroups(torch.autograd.Function):
    """``t -> (t[:, 0:c], t[:, c:2c], ...)`` as views; the backward hands back the shared gradient buffer when every slice gradient it receives IS
    that buffer's slice (what ``_SpadeModulate`` writes), else it assembles the gradient the ordinary way."""
    @staticmethod
    def forward(ctx, t, holder):
        ctx.holder = holder
        ctx.set_materialize_grads(False)
        c = t.shape[1] // holder.groups
        return tuple(t.narrow(1, g * c, c) for g in range(holder.groups))

    @staticmethod
    def backward(ctx, *grads):
        # the holder stays with the node (a second backward pass over a retained graph finds it again); the BUFFER is handed out once and the
        # holder forgets it, so the next pass writes a fresh one instead of a tensor someone already holds as a gradient
        h = ctx.holder
        buf, row = h.buf, h.row
        if buf is not None and all(g is not None and g.data_ptr() == h.view(i).data_ptr() and g.shape == h.view(i).shape and g.stride() == h.view(i).stride()
                                   for i, g in enumerate(grads)):
            h.buf = h.row = None
            return _native.amax_attach(buf, row), None
        h.buf = h.row = None
        c = h.shape[1] // h.groups
        parts = [g if g is not None else torch.zeros([h.shape[0], c, *h.shape[2:]], dtype=h.dtype, device=h.device) for g in grads]
        return torch.cat(parts, dim=1), None

def spade_modulate(x, gamma, beta, eps=1e-5, relu_gain=None, clamp=None, shared=None, passthrough=False):
    """InstanceNorm(x) * (1 + gamma) + beta; ``relu_gain`` not None additionally applies ``min(relu(.) * relu_gain, clamp)``
    in the same pass (the activation of the Spade_Conv2dLayer that consumes the result).  ``beta=None``: ``gamma`` holds
    gamma | beta as the two channel halves of a [N, 2C, H, W] tensor."""
    _native.require_gpu(x, 'spade_modulate')
    if x.dtype not in _HIP_DTYPES or gamma.dtype != x.dtype or (beta is not None and beta.dtype != x.dtype):
        raise RuntimeError('spade_modulate: x, gamma and beta must share one of float32 / float16 / bfloat16')
    post = (0, 1.0, -1.0) if relu_gain is None else (2, float(relu_gain), float(clamp if clamp is not None else -1))
    if passthrough:         # (out, x'): x' = x for the other normalisation of the same tensor (_SpadeModulate.forward)
        if torch.is_grad_enabled() and x.requires_grad:
            return _SpadeModulate.apply(x, gamma, beta, eps, post, shared, True)
        return _SpadeModulate.apply(x, gamma, beta, eps, post, shared), x
    return _SpadeModulate.apply(x, gamma, beta, eps, post, shared)

#----------------------------------------------------------------------------

import os as _os
_GARMENT_FUSED = _os.environ.get('PASTA_GARMENT_FUSED', '1') != '0'         # A/B switch: 0 = the reference's element-wise passes + torch.cat

class _GarmentFeat(torch.autograd.Function):
    """``cat([fill(feat_u), fill(feat_l)], dim=1)`` with ``fill(x) = x * (1 - hole) + (sum_hw(x * valid) / count) * hole`` -- the tail of
    ``get_spade_feat`` for the upper and the lower garment (networks.py:5777-5800, 5836) -- by ``pasta_masked_mean_fill``: one workgroup
    per (n, c) plane holds the plane in registers between the sum and the fill and writes into its half of the result."""
    @staticmethod
    def forward(ctx, feat_u, valid_u, hole_u, count_u, feat_l, valid_l, hole_l, count_l):
        n, c, h, w = feat_u.shape
        out = torch.empty([n, 2 * c, h, w], dtype=torch.float32, device=feat_u.device)
        row = _native.amax_slot(out)
        keep = []
        lib = _native.lib()
        with torch.cuda.device(out.device):
            for k, (f, v, ho, cnt) in enumerate(((feat_u, valid_u, hole_u, count_u), (feat_l, valid_l, hole_l, count_l))):
                f, v, ho = f.contiguous(), v.reshape(n, h * w).contiguous(), ho.reshape(n, h * w).contiguous()
                inv = (1.0 / cnt.reshape(n).float()).contiguous()
                _native.check(lib.pasta_masked_mean_fill(_native.ptr(f), _native.ptr(v), _native.ptr(ho), _native.ptr(inv),
                                                         ctypes.c_void_p(out.data_ptr() + 4 * k * c * h * w), n, c, h * w, 0, 2 * c * h * w, 0,
                                                         _native.stream(), _native.ptr(row)))
                keep += [v, ho, inv]
        _native.amax_attach(out, row)
        ctx.save_for_backward(*keep)
        ctx.dims = (n, c, h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        n, c, h, w = ctx.dims
        saved = ctx.saved_tensors
        grads = []
        if torch.is_grad_enabled() and dout.requires_grad:          # a gradient of this gradient is wanted: differentiable torch operations
            for k in range(2):
                v, ho, inv = (t.reshape(n, 1, h, w) if t.ndim == 2 else t.reshape(n, 1, 1, 1) for t in saved[3 * k:3 * k + 3])
                d = dout[:, k * c:(k + 1) * c]
                grads.append(d * (1 - ho) + v * inv * (d * ho).sum(dim=(2, 3), keepdim=True))
        else:
            dout = dout.contiguous()
            lib = _native.lib()
            with torch.cuda.device(dout.device):
                for k in range(2):
                    v, ho, inv = saved[3 * k:3 * k + 3]
                    dx = torch.empty([n, c, h, w], dtype=torch.float32, device=dout.device)
                    row = _native.amax_slot(dx) if ctx.needs_input_grad[4 * k] else None
                    if ctx.needs_input_grad[4 * k]:
                        _native.check(lib.pasta_masked_mean_fill(ctypes.c_void_p(dout.data_ptr() + 4 * k * c * h * w), _native.ptr(v), _native.ptr(ho),
                                                                 _native.ptr(inv), _native.ptr(dx), n, c, h * w, 2 * c * h * w, 0, 1, _native.stream(),
                                                                 _native.ptr(row)))
                        _native.amax_attach(dx, row)
                    grads.append(dx if ctx.needs_input_grad[4 * k] else None)
        return grads[0], None, None, None, grads[1], None, None, None

_MBA_AMAX = _os.environ.get('PASTA_MBA_AMAX', '1') != '0'         # A/B switch: 0 = the consumer scans the output of mod_bias_act

class _ModBiasAct(torch.autograd.Function):
    """Tail of a modulated-convolution layer in one pass: ``clamp(act(u * d[n,c] + noise * strength + b[c]) * gain)``
    (the demodulation + noise of modulated_conv2d, networks.py:77-82, and SynthesisLayer's bias_act, :313-314).
    The backward pass is one kernel as well: du and, per plane, the sums that give dd, dstrength and db."""
    @staticmethod
    def forward(ctx, u, d, noise, strength, b, cfg):
        act_idx, alpha, gain, clamp = cfg
        n, c, h, w = u.shape
        u = u.contiguous()
        d = d.contiguous() if d is not None else None
        per_sample = 0
        if noise is not None:
            per_sample = int(noise.numel() == n * h * w and n > 1)
            assert noise.numel() in (h * w, n * h * w)
            noise = noise.contiguous()
        y = torch.empty_like(u)
        # producer maxima for the large planes (one commit per 4096-element workgroup, most of them skipped by the look at the slot): the layer's
        # output is the operand of the next modulated convolution, whose launch would scan it (round 4: 150.35 -> 150.0 ms same box; small
        # tensors keep the scan: profiles/r4_ab_mba_amax.txt)
        row = _native.amax_slot(y) if (_MBA_AMAX and y.numel() >= 1 << 22) else None
        with torch.cuda.device(u.device):
            st = _native.lib().pasta_mod_bias_act(_native.ptr(u), _native.ptr(d), _native.ptr(noise), _native.ptr(strength), _native.ptr(b),
                                                  _native.ptr(y), _native.dtype_code(u, 'mod_bias_act'), n, c, h * w, per_sample, act_idx, float(alpha), float(gain), float(clamp),
                                                  _native.stream(), _native.ptr(row))
        _native.check(st)
        _native.amax_attach(y, row)
        ctx.save_for_backward(u, d, noise, y)
        ctx.cfg, ctx.per_sample = cfg, per_sample
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        u, d, noise, y = ctx.saved_tensors
        act_idx, alpha, gain, clamp = ctx.cfg
        n, c, h, w = u.shape
        dy = dy.contiguous()
        lib = _native.lib()
        du = torch.empty_like(u)
        row = _native.amax_slot(du)
        part = torch.empty([lib.pasta_mod_bias_act_bwd_workspace(n, c, h * w) // 4], dtype=torch.float32, device=u.device)
        with torch.cuda.device(u.device):
            st = lib.pasta_mod_bias_act_bwd(_native.ptr(dy), _native.ptr(y), _native.ptr(u), _native.ptr(d), _native.ptr(noise), _native.ptr(du),
                                            _native.ptr(part), _native.dtype_code(u, 'mod_bias_act_bwd'), n, c, h * w, ctx.per_sample, act_idx, float(alpha), float(gain), float(clamp),
                                            _native.stream(), _native.ptr(row))
        _native.check(st)
        _native.amax_attach(du, row)
        sums = part.reshape(n, c, -1, 3).sum(dim=2)                    # [N, C, 3]: sum dz*u, sum dz*noise, sum dz
        dd = sums[:, :, 0] if d is not None and ctx.needs_input_grad[1] else None
        dstrength = sums[:, :, 1].sum() if noise is not None and ctx.needs_input_grad[3] else None
        db = sums[:, :, 2].sum(dim=0) if ctx.needs_input_grad[4] else None
        return du, dd, None, dstrength, db, None

def mod_bias_act(u, dcoefs, noise, strength, bias, act='lrelu', alpha=None, gain=None, clamp=None):
    """``bias_act(u * dcoefs[n,c] + noise * strength, bias, act, alpha, gain, clamp)`` in one pass; ``noise`` is the
    unit noise ([H,W] or [N,1,H,W]) and ``strength`` the 0-dim strength parameter (both None = no noise).
    First-order differentiable (the generator takes no double backward)."""
    _native.require_gpu(u, 'mod_bias_act')
    spec = bias_act.activation_funcs[act]
    assert act in ('linear', 'lrelu') and u.dtype in _HIP_DTYPES
    cfg = (spec.cuda_idx, float(alpha if alpha is not None else spec.def_alpha), float(gain if gain is not None else spec.def_gain),
           float(clamp if clamp is not None else -1))
    return _ModBiasAct.apply(u, dcoefs, noise, strength, bias, cfg)

#----------------------------------------------------------------------------

#----------------------------------------------------------------------------
# Helpers shared by the layer classes.

def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """x / rms(x) along ``dim`` (networks.py:30-32)."""
    return x * torch.rsqrt(x.square().mean(dim=dim, keepdim=True) + eps)

def _attach_filter(module, taps):
    module.register_buffer('resample_filter', upfirdn2d.setup_filter(taps))

def _fresh_weight(out_channels, in_channels, kernel_size, channels_last=False):
    w = torch.randn([out_channels, in_channels, kernel_size, kernel_size])
    return w.to(memory_format=torch.channels_last) if channels_last else w

def _fan_in_gain(in_channels, kernel_size):
    return 1 / np.sqrt(in_channels * kernel_size * kernel_size)

def _scaled_act(activation, gain, conv_clamp):
    """Activation gain and clamp of a layer invoked with an extra ``gain``; the clamp scales with it (networks.py:176-177)."""
    return bias_act.activation_funcs[activation].def_gain * gain, (None if conv_clamp is None else conv_clamp * gain)

def _hip_act(x):
    """Is ``x`` an activation tensor the fused HIP kernels take (fp32, or 16-bit storage: fp16 / bf16)?"""
    return x.dtype in _HIP_DTYPES and x.device.type == 'cuda'

def _master_weight(weight, x):
    """The weight operand for a convolution over ``x``: 16-bit activations on the GPU meet the fp32 master directly (the
    packing kernel of the convolution rounds it to the operand type -- what ``weight.to(x.dtype)`` does in the reference,
    networks.py:171, without a cast kernel and without a 16-bit weight-gradient detour); otherwise ``weight.to(x.dtype)``."""
    return weight if (x.device.type == 'cuda' and x.dtype != torch.float32) else weight.to(x.dtype)

def _as_dtype(name):
    """'float16' / 'bfloat16' / torch dtype / None -> torch dtype or None"""
    if name is None or isinstance(name, torch.dtype):
        return name
    dtype = getattr(torch, str(name))
    assert dtype in (torch.float16, torch.bfloat16, torch.float32)
    return None if dtype == torch.float32 else dtype

def _block_dtype(use_fp16, channels_last, force_fp32, half_dtype=torch.float16):
    half = use_fp16 and not force_fp32
    return (half_dtype if half else torch.float32), (torch.channels_last if channels_last and not force_fp32 else torch.contiguous_format)

#----------------------------------------------------------------------------
# Modulated convolution.

def _demodulation(weight, styles):
    """rsqrt(sum_{i,kh,kw} (w[o,i] s[n,i])^2 + 1e-8) [N, O]: the per-sample weight tensor [N,O,I,k,k] of networks.py:65-68 is
    never formed.  On the GPU one kernel (``pasta_demod_coefs``: a workgroup per output channel, wavefront-shuffle
    reduction over the input channels); elsewhere a matrix product of the squared styles with the tap-summed squared weights."""
    if weight.device.type == 'cuda' and weight.dtype == torch.float32 and styles.dtype == torch.float32 and weight.shape[1] <= 4096:
        return conv2d_gradfix.demod_coefs(weight, styles)
    return torch.rsqrt(styles.square() @ weight.square().sum(dim=[2, 3]).t() + 1e-8)

def _forward_only(*tensors):
    """No graph will be recorded for an operation on these tensors: the one-launch forward kernels may run."""
    return not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors))

def _styled_conv_forward(x, weight, styles, *, up=1, padding=0, resample_filter=None, flip_weight=True, demodulate=True, per_sample=False,
                         noise=None, strength=None, bias=None, act='linear', gain=None, clamp=None):
    """Forward-only SynthesisLayer / ToRGBLayer body (networks.py:36-94, 302-314): modulation in the convolution's activation
    staging (shared weight) or weight packing (``per_sample``: the reference's fused_modconv form); demodulation, noise,
    bias, activation and clamp in its epilogue.  An upsampling layer filters between the two, so its tail is one pass of
    ``mod_bias_act`` after the filter.  ``noise`` = unit-variance plane(s) before the learnt ``strength``."""
    if x.dtype == torch.float16 and demodulate:           # networks.py:57-59
        weight = weight * (_fan_in_gain(weight.shape[1], np.sqrt(weight[0, 0].numel())) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)
    dcoefs = _demodulation(weight, styles) if demodulate else None
    if up == 1:
        w = weight if flip_weight else weight.flip([2, 3])
        tail = dict(noise=noise, strength=strength, bias=bias, act=act, gain=gain, clamp=clamp)
        return conv2d_gradfix.modulated_conv2d_forward(x, w, styles, dcoefs, padding=padding, per_sample=per_sample, tail=tail)
    u = conv2d_resample.conv2d_resample(x=x, w=weight, f=resample_filter, up=up, padding=padding, flip_weight=flip_weight,
                                        modulation=(styles, dcoefs, per_sample))
    with torch.no_grad():
        return mod_bias_act(u, (None if per_sample else dcoefs), noise, strength, bias, act=act, gain=gain, clamp=clamp)

def _modulate_and_convolve(x, weight, styles, up, down, padding, resample_filter, demodulate, flip_weight):
    """Shared-weight form of the modulated convolution up to, not including, the demodulation (networks.py:72-76):
    returns conv(x * s) and the demodulation coefficients (None without demodulation)."""
    dcoefs = _demodulation(weight, styles) if demodulate else None
    w = _master_weight(weight, x)
    if up == 1 and down == 1 and isinstance(padding, int) and (flip_weight or tuple(w.shape[2:]) == (1, 1)) and \
            conv2d_gradfix.modconv_available(x, w, styles, padding=padding):
        # the styles ride in the staging of the forward launch, in the epilogue of the input gradient and in the reduction of the weight
        # gradient: x * styles is never formed (conv2d_gradfix.modulated_conv2d_shared)
        return conv2d_gradfix.modulated_conv2d_shared(x, w, styles, padding=padding), dcoefs
    y = conv2d_resample.conv2d_resample(x=scale_planes(x, styles), w=w, f=resample_filter, up=up, down=down,
                                        padding=padding, flip_weight=flip_weight)
    return y, dcoefs

def _per_sample_convolve(x, weight, styles, up, down, padding, resample_filter, demodulate, flip_weight):
    """Per-sample-weight form (networks.py:84-94): the batch becomes the group dimension of one grouped convolution."""
    n = int(x.shape[0])
    o, i, kh, kw = weight.shape
    w = weight[None] * styles[:, None, :, None, None]                               # [N, O, I, kh, kw]
    if demodulate:
        w = w * torch.rsqrt(w.square().sum(dim=[2, 3, 4], keepdim=True) + 1e-8)
    y = conv2d_resample.conv2d_resample(x=x.reshape(1, n * i, *x.shape[2:]), w=w.reshape(n * o, i, kh, kw).to(x.dtype),
                                        f=resample_filter, up=up, down=down, padding=padding, groups=n, flip_weight=flip_weight)
    return y.reshape(n, o, *y.shape[2:])

def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=True, fused_modconv=True):
    """StyleGAN2 modulated convolution (networks.py:36-94).  ``x`` [N,I,H,W], ``weight`` [O,I,kh,kw], ``styles`` [N,I],
    ``noise`` broadcastable to the output or None; ``padding`` is relative to the upsampled image.

    ``fused_modconv=False`` (training): activations are scaled by the styles, ONE shared-weight convolution runs for the
    whole batch, and its output is scaled by the demodulation coefficients (+ noise).
    ``fused_modconv=True`` (inference): per-sample weights, one grouped convolution."""
    n = x.shape[0]
    o, i, kh, kw = weight.shape
    misc.assert_shape(x, [n, i, None, None])
    misc.assert_shape(styles, [n, i])
    if x.dtype == torch.float16 and demodulate:
        # fp16 cannot hold the products of raw weights and styles; demodulation cancels any common factor, so both are
        # brought to unit max-norm first (networks.py:57-59)
        weight = weight * (_fan_in_gain(i, np.sqrt(kh * kw)) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)
    if fused_modconv and not (_hip_act(x) and o < 128):
        y = _per_sample_convolve(x, weight, styles, up, down, padding, resample_filter, demodulate, flip_weight)
        return y if noise is None else y.add_(noise)
    # (fused_modconv with fewer than 128 output channels on the GPU: the N groups of the per-sample-weight form would each
    # fill a quarter or less of a 128-row matrix tile -- 45..73 TFLOP/s measured on the 32- and 64-channel layers of the 512
    # generator -- so the algebraically identical shared-weight form below runs instead; same result to fp32 rounding)
    y, dcoefs = _modulate_and_convolve(x, weight, styles, up, down, padding, resample_filter, demodulate, flip_weight)
    if dcoefs is None:
        return y if noise is None else y.add_(noise.to(y.dtype))
    if noise is None:
        return scale_planes(y, dcoefs)
    return fma.fma(y, dcoefs.to(y.dtype).reshape(n, -1, 1, 1), noise.to(y.dtype))

#----------------------------------------------------------------------------
# Dense layers.

@persistence.persistent_class
class FullyConnectedLayer(torch.nn.Module):
    """y = act(x @ (w * lr_multiplier / sqrt(in)).T + b * lr_multiplier) (networks.py:98-128)."""
    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        # both gains ride in the GEMM's alpha / beta: no scaling kernels, forwards (addmm) or backwards (_ScaledLinear)
        w = self.weight.to(x.dtype)
        b = None if self.bias is None else self.bias.to(x.dtype)
        if _FC_FUSED_GRADS and x.ndim == 2 and x.device.type == 'cuda' and x.dtype == torch.float32 and x.shape[0] > 0:
            y = _ScaledLinear.apply(x, w, b, float(self.weight_gain), float(self.bias_gain))
        elif b is None:
            y = torch.mm(x, w.t()) * self.weight_gain
        else:
            y = torch.addmm(b[None], x, w.t(), beta=float(self.bias_gain), alpha=float(self.weight_gain))
        return y if self.activation == 'linear' else bias_act.bias_act(y, None, act=self.activation)

# A/B switch: 0 = autograd's own backward of addmm (a GEMM and a scaling kernel per gradient, a sum and a scaling kernel for the bias)
_FC_FUSED_GRADS = _os.environ.get('PASTA_FC_FUSED_GRADS', '1') != '0'

class _ScaledLinear(torch.autograd.Function):
    """``y = alpha x w^T + beta b`` (FullyConnectedLayer: reference networks.py:117-128 with the gains folded into the GEMM) whose backward is three
    launches -- ``dx = alpha dy w``, ``dw = alpha dy^T x``, ``db = beta dy^T 1`` as GEMM / GEMV calls that carry their factor -- where autograd's
    backward of ``addmm`` with ``alpha`` / ``beta`` runs six (a product and a scaling kernel each, a sum and a scaling kernel for the bias):
    about 80 dense layers per training step (the affine layer of every synthesis layer, the mapping networks).  The backward is written with
    differentiable operators, so gradients of any order follow (R1 passes through the discriminator's dense layers)."""
    @staticmethod
    def forward(ctx, x, w, b, alpha, beta):
        ctx.save_for_backward(x, w)
        ctx.alpha, ctx.beta, ctx.has_b = alpha, beta, b is not None
        if b is None:
            return torch.addmm(x[:1, :1], x, w.t(), beta=0.0, alpha=alpha)       # beta = 0: the addend is ignored (a view: no kernel to make one)
        return torch.addmm(b[None], x, w.t(), beta=beta, alpha=alpha)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        ignored = dy[:1, :1]                        # the addend of a product with beta = 0 (never read; NaN / inf in it do not propagate)
        dx = torch.addmm(ignored, dy, w, beta=0.0, alpha=ctx.alpha) if ctx.needs_input_grad[0] else None
        dw = torch.addmm(ignored, dy.t(), x, beta=0.0, alpha=ctx.alpha) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = torch.addmv(ignored[0], dy.t(), _ones_vector(dy), beta=0.0, alpha=ctx.beta)
        return dx, dw, db, None, None

_ones_cache = {}

def _ones_vector(like):
    """[N] ones on ``like``'s device and of its dtype, one per (device, dtype, N) for the life of the process (the summing vector of the bias gradient)."""
    key = (like.device, like.dtype, int(like.shape[0]))
    v = _ones_cache.get(key)
    if v is None:
        v = _ones_cache[key] = torch.ones([like.shape[0]], dtype=like.dtype, device=like.device)
    return v

@persistence.persistent_class
class MappingNetwork(torch.nn.Module):
    """(z, embed(c)) -> FC stack -> w, broadcast to ``num_ws`` rows; keeps the running mean ``w_avg`` (networks.py:183-259)."""
    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.995):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.num_ws, self.num_layers, self.w_avg_beta = num_ws, num_layers, w_avg_beta
        embed = 0 if c_dim == 0 else (w_dim if embed_features is None else embed_features)
        hidden = w_dim if layer_features is None else layer_features
        widths = [z_dim + embed] + [hidden] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed)
        for idx, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            setattr(self, f'fc{idx}', FullyConnectedLayer(fan_in, fan_out, activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False):
        parts = []
        if self.z_dim > 0:
            misc.assert_shape(z, [None, self.z_dim])
            parts.append(normalize_2nd_moment(z.to(torch.float32)))
        if self.c_dim > 0:
            misc.assert_shape(c, [None, self.c_dim])
            parts.append(normalize_2nd_moment(self.embed(c.to(torch.float32))))
        x = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x)
        if self.training and self.w_avg_beta is not None and not skip_w_avg_update:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x[:, None, :].repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            assert self.w_avg_beta is not None
            rows = slice(None) if (self.num_ws is None or truncation_cutoff is None) else slice(0, truncation_cutoff)
            if self.num_ws is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, rows] = self.w_avg.lerp(x[:, rows], truncation_psi)
        return x

#----------------------------------------------------------------------------
# Convolution layers with a fixed (non-modulated) weight.

class _FilteredConv(torch.nn.Module):
    """State shared by ``Conv2dLayer`` (convolution then activation) and ``Spade_Conv2dLayer`` (activation then
    convolution): weight / bias as parameters or, frozen, as buffers (networks.py:158-168), the resampling filter, and
    the equalised-learning-rate gain, which the convolution's weight packing applies (no kernel for ``w * gain``)."""
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=[1,3,3,1], conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        self.activation, self.up, self.down, self.conv_clamp = activation, up, down, conv_clamp
        _attach_filter(self, resample_filter)
        self.padding = kernel_size // 2
        self.weight_gain = _fan_in_gain(in_channels, kernel_size)
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        weight = _fresh_weight(out_channels, in_channels, kernel_size, channels_last)
        bias = torch.zeros([out_channels]) if bias else None
        if trainable:
            self.weight = torch.nn.Parameter(weight)
            self.bias = torch.nn.Parameter(bias) if bias is not None else None
        else:
            self.register_buffer('weight', weight)
            if bias is None:
                self.bias = None
            else:
                self.register_buffer('bias', bias)

    def _resample_args(self, x):
        return dict(w=_master_weight(self.weight, x), f=self.resample_filter, up=self.up, down=self.down, padding=self.padding,
                    flip_weight=(self.up == 1), wgain=self.weight_gain)     # up: the transposed convolution wants true-convolution taps

@persistence.persistent_class
class Conv2dLayer(_FilteredConv):
    """conv2d_resample -> bias_act, bias / activation / gain / clamp in the convolution's epilogue where the dense
    convolution is the last step (networks.py:132-179)."""
    def forward(self, x, gain=1, passthrough=False, add=None):
        """``passthrough=True`` (own extension) returns ``(y, x')``: hand ``x'`` instead of ``x`` to the other consumers of ``x`` (a residual
        block's skip branch) and their gradient is added in the epilogue of this layer's input-gradient launch (conv2d_gradfix._ConvBiasActHip).
        ``add`` (own extension; shape of the output): returns ``layer(x, gain) + add`` -- for a linear, bias-free, unclamped layer on fp32 GPU
        tensors (the skip branch of the residual blocks) the sum is formed in the convolution's epilogue, the gain folded into the weight
        gain; otherwise by ``add_`` as the reference writes it (networks.py:557, :994)."""
        if add is not None:
            if _SKIP_ADD_FUSED and self.activation == 'linear' and self.bias is None and self.conv_clamp is None and x.dtype == torch.float32 \
                    and x.device.type == 'cuda' and self.up == 1 and not passthrough:
                args = self._resample_args(x)
                args['wgain'] = args['wgain'] * float(gain)
                return conv2d_resample.conv2d_resample_bias_act(x=x, b=None, act='linear', gain=1, clamp=None, residual=add, **args)
            y = self.forward(x, gain=gain, passthrough=passthrough)
            if passthrough:                             # (y, x'): the sum belongs to the first element
                return y[0].add_(add), y[1]
            return y.add_(add)
        act_gain, act_clamp = _scaled_act(self.activation, gain, self.conv_clamp)
        return conv2d_resample.conv2d_resample_bias_act(x=x, b=(None if self.bias is None else self.bias.to(x.dtype)), act=self.activation,
                                                        gain=act_gain, clamp=act_clamp, passthrough=passthrough, **self._resample_args(x))

# A/B switch: 0 = ``shortcut.add_(...)`` as the reference writes the residual blocks; 1 = the sum in the skip convolution's epilogue
_SKIP_ADD_FUSED = _os.environ.get('PASTA_SKIP_ADD_FUSED', '1') != '0'

@persistence.persistent_class
class Spade_Conv2dLayer(_FilteredConv):
    """bias_act first, convolution second (networks.py:4304-4355).  ``no_act=True``: the caller already applied the
    activation (the SPADE normalisation kernel does it in the same pass)."""
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='relu', up=1, down=1,
                 resample_filter=[1,3,3,1], conv_clamp=None, channels_last=False, trainable=True):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, activation=activation, up=up, down=down,
                         resample_filter=resample_filter, conv_clamp=conv_clamp, channels_last=channels_last, trainable=trainable)

    def fusable_activation(self, gain=1):
        """(relu gain, clamp) when the producer of this layer's input may apply its activation (bias-free relu), else None."""
        if self.bias is None and self.activation == 'relu':
            return _scaled_act(self.activation, gain, self.conv_clamp)
        return None

    def forward(self, x, gain=1, no_act=False, residual=None):
        """``residual`` (shape of the output): added in the convolution's epilogue -- the block's ``shortcut.add_(x)`` without its
        pass over HBM (two reads and a write of the activation)."""
        if not no_act:
            act_gain, act_clamp = _scaled_act(self.activation, gain, self.conv_clamp)
            x = bias_act.bias_act(x, (None if self.bias is None else self.bias.to(x.dtype)), act=self.activation, gain=act_gain, clamp=act_clamp)
        if residual is not None:
            return conv2d_resample.conv2d_resample_bias_act(x=x, b=None, act='linear', gain=1, residual=residual, **self._resample_args(x))
        return conv2d_resample.conv2d_resample(x=x, **self._resample_args(x))

#----------------------------------------------------------------------------
# Style-modulated layers.

class _StyledConv(torch.nn.Module):
    """affine (w -> per-channel styles, bias initialised to 1), a weight and a bias."""
    def _make_styled(self, in_channels, out_channels, w_dim, kernel_size, channels_last):
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(_fresh_weight(out_channels, in_channels, kernel_size, channels_last))

@persistence.persistent_class
class SynthesisLayer(_StyledConv):
    """affine -> modulated 3x3 convolution (optionally x2) + noise -> bias, lrelu, clamp (networks.py:263-315)."""
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True, activation='lrelu',
                 resample_filter=[1,3,3,1], conv_clamp=None, channels_last=False):
        super().__init__()
        self.resolution, self.up, self.use_noise = resolution, up, use_noise
        self.activation, self.conv_clamp = activation, conv_clamp
        _attach_filter(self, resample_filter)
        self.padding = kernel_size // 2
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self._make_styled(in_channels, out_channels, w_dim, kernel_size, channels_last)
        if use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def _unit_noise(self, x, noise_mode):
        """Unit-variance noise plane(s) for this call, before the learnt strength; None when the layer adds none."""
        if not self.use_noise or noise_mode == 'none':
            return None
        if noise_mode == 'const':
            return self.noise_const
        return torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device)

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1):
        assert noise_mode in ['random', 'const', 'none']
        side = self.resolution // self.up
        misc.assert_shape(x, [None, self.weight.shape[1], side, side])
        styles = self.affine(w)
        unit = self._unit_noise(x, noise_mode)
        act_gain, act_clamp = _scaled_act(self.activation, gain, self.conv_clamp)
        conv = dict(up=self.up, padding=self.padding, resample_filter=self.resample_filter, flip_weight=(self.up == 1))
        if _hip_act(x) and self.activation in ('linear', 'lrelu') and _forward_only(x, styles, self.weight, self.bias) and \
                not (fused_modconv and unit is not None and unit.ndim == 4 and self.weight.shape[0] >= 128):
            # no graph: the whole layer is the convolution launch (+ the filter and one tail pass when upsampling)
            return _styled_conv_forward(x, self.weight, styles, **conv, per_sample=(fused_modconv and self.weight.shape[0] >= 128), noise=unit,
                                        strength=(None if unit is None else self.noise_strength), bias=self.bias, act=self.activation,
                                        gain=act_gain, clamp=act_clamp)
        if not fused_modconv and _hip_act(x) and self.activation in ('linear', 'lrelu'):
            # training: demodulation, noise, bias, activation and clamp are ONE pass over the convolution's output
            u, dcoefs = _modulate_and_convolve(x, self.weight, styles, down=1, demodulate=True, **conv)
            return mod_bias_act(u, dcoefs, unit, (None if unit is None else self.noise_strength), self.bias, act=self.activation,
                                gain=act_gain, clamp=act_clamp)
        noise = None if unit is None else unit * self.noise_strength
        y = modulated_conv2d(x=x, weight=self.weight, styles=styles, noise=noise, fused_modconv=fused_modconv, **conv)
        return bias_act.bias_act(y, self.bias.to(y.dtype), act=self.activation, gain=act_gain, clamp=act_clamp)

class _StyledHeads(_StyledConv):
    """1x1 modulated convolution without demodulation -> bias -> clamp, for the image and for any extra per-pixel heads
    that read the same styled activations.  ``heads`` = ((suffix, channels, activation), ...) creates ``m_weight<suffix>``
    / ``m_bias<suffix>``.  All linear outputs (image + linear heads) come from ONE convolution over the concatenated
    weights: the activations (268 MB at 256^2, batch 16) are scaled and read once instead of once per head."""
    def _make_heads(self, in_channels, out_channels, w_dim, kernel_size, conv_clamp, channels_last, heads=()):
        self.conv_clamp = conv_clamp
        self._make_styled(in_channels, out_channels, w_dim, kernel_size, channels_last)
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = _fan_in_gain(in_channels, kernel_size)
        self._heads = tuple(heads)
        for suffix, channels, _act in self._heads:
            setattr(self, f'm_weight{suffix}', torch.nn.Parameter(_fresh_weight(channels, in_channels, kernel_size, channels_last)))
            setattr(self, f'm_bias{suffix}', torch.nn.Parameter(torch.zeros([channels])))

    def _project(self, x, w, fused_modconv):
        """-> (image, [head outputs in declaration order])"""
        styles = self.affine(w) * self.weight_gain
        def run(weight, bias, act):
            if _hip_act(x) and act in conv2d_gradfix.FUSABLE_ACTS and _forward_only(x, styles, weight, bias):
                return _styled_conv_forward(x, weight, styles, demodulate=False, bias=bias, act=act, clamp=self.conv_clamp)
            y = modulated_conv2d(x=x, weight=weight, styles=styles, demodulate=False, fused_modconv=fused_modconv)
            return bias_act.bias_act(y, bias.to(y.dtype), act=act, clamp=self.conv_clamp)
        linear = [s for s, _c, act in self._heads if act == 'linear']
        weights = [self.weight] + [getattr(self, f'm_weight{s}') for s in linear]
        biases = [self.bias] + [getattr(self, f'm_bias{s}') for s in linear]
        if len(weights) == 1:
            merged = [run(self.weight, self.bias, 'linear')]
        else:
            merged = run(torch.cat(weights, dim=0), torch.cat(biases, dim=0), 'linear').split([wt.shape[0] for wt in weights], dim=1)
        by_suffix = dict(zip(linear, merged[1:]))
        for s, _c, act in self._heads:
            if act != 'linear':
                by_suffix[s] = run(getattr(self, f'm_weight{s}'), getattr(self, f'm_bias{s}'), act)
        return merged[0], [by_suffix[s] for s, _c, _a in self._heads]

@persistence.persistent_class
class ToRGBLayer(_StyledHeads):
    """networks.py:319-334"""
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        self._make_heads(in_channels, out_channels, w_dim, kernel_size, conv_clamp, channels_last)

    def forward(self, x, w, fused_modconv=True):
        return self._project(x, w, fused_modconv)[0]

@persistence.persistent_class
class ToRGBLayerFull(_StyledHeads):
    """ToRGB whose last style-branch instance also emits the 6-class parsing logits (networks.py:5582-5611).
    Returns (image, parsing logits or None)."""
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False, is_last=False, is_style=False):
        super().__init__()
        self.is_last, self.is_style = is_last, is_style
        self._make_heads(in_channels, out_channels, w_dim, kernel_size, conv_clamp, channels_last,
                         heads=[('1', 6, 'linear')] if (is_last and is_style) else [])

    def forward(self, x, w, fused_modconv=True):
        img, heads = self._project(x, w, fused_modconv)
        return img, (heads[0] if heads else None)

@persistence.persistent_class
class ToRGBLayerV18(_StyledHeads):
    """ToRGB of the released 256x192 inference model: its last instance also emits sigmoid upper / lower clothing masks
    (networks.py:5276-5310).  Returns (image, upper mask or None, lower mask or None)."""
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False, is_last=False):
        super().__init__()
        self.is_last = is_last
        self._make_heads(in_channels, out_channels, w_dim, kernel_size, conv_clamp, channels_last,
                         heads=[('1', 1, 'sigmoid'), ('2', 1, 'sigmoid')] if is_last else [])

    def forward(self, x, w, fused_modconv=True):
        img, heads = self._project(x, w, fused_modconv)
        return (img, *heads) if heads else (img, None, None)

#----------------------------------------------------------------------------
# Encoders.

# A/B switch: 0 = every consumer of a multi-consumer tensor returns its own input gradient and autograd adds them (two reads and a write per addition)
_GRAD_JOIN = _os.environ.get('PASTA_GRAD_JOIN', '1') != '0'

def _layer_and_input(layer, x):
    """``(layer(x), x')`` for a tensor with further consumers: give them ``x'`` (= ``x``), and their gradient is added in the epilogue of
    the layer's input-gradient launch instead of by a pass of its own (``Conv2dLayer.forward(passthrough=True)``)."""
    # fp32 storage only: in 16-bit storage the reference rounds each consumer's gradient to the storage type BEFORE the addition, and the
    # config-5 fixtures (tests/test_config5_gpu.py) hold the path to that sequence of roundings
    return layer(x, passthrough=True) if (_GRAD_JOIN and x.dtype == torch.float32) else (layer(x), x)

@persistence.persistent_class
class ResBlock(torch.nn.Module):
    """sqrt(1/2) * (1x1 skip + 3x3 -> 3x3), resampling in skip and first 3x3 (networks.py:528-558).  ``kernel_size`` is
    accepted and ignored, as in the reference (its callers pass 4)."""
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=[1,3,3,1], conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        _attach_filter(self, resample_filter)
        shared = dict(resample_filter=resample_filter, conv_clamp=conv_clamp, channels_last=channels_last)
        self.conv0 = Conv2dLayer(in_channels, out_channels, kernel_size=3, activation=activation, up=up, down=down, bias=bias, **shared)
        self.conv1 = Conv2dLayer(out_channels, out_channels, kernel_size=3, activation=activation, bias=bias, **shared)
        self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=up, down=down, **shared)

    def forward(self, x):
        half = np.sqrt(0.5)
        h, x = _layer_and_input(self.conv0, x)          # x again: the skip branch's gradient joins conv0's input gradient in that launch
        return self.skip(x, gain=half, add=self.conv1(h, gain=half))      # skip(x) + conv1(.): the sum in the skip convolution's epilogue

# channel multipliers (in, out) of the pose encoder's stride-2 stages (networks.py:564-565); the table continues at
# 8 -> 8 for pyramids deeper than the reference's six stages (512^2 and up, own generalisation)
_POSE_STAGES = [(1, 2), (2, 4), (4, 4), (4, 4), (4, 8), (8, 8)]

@persistence.persistent_class
class ConstEncoderNetwork(nn.Module):
    """Pose (+ retained image) encoder: 1x1 stem, then ``n_downsampling`` stride-2 3x3 layers (networks.py:560-579)."""
    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=4):
        super().__init__()
        stages = (_POSE_STAGES + [(8, 8)] * n_downsampling)[:n_downsampling]
        self.model = nn.Sequential(Conv2dLayer(input_nc, ngf, kernel_size=1),
                                   *[Conv2dLayer(ngf * a, ngf * b, kernel_size=3, down=2) for a, b in stages])

    def forward(self, x):
        return self.model(x)

class Dense(nn.Module):
    """Per-pixel Linear -> InstanceNorm -> LeakyReLU(0.01) (networks.py:594-611).  The Linear runs as a 1x1 convolution on
    the NCHW tensor (bias in its epilogue), so the reference's two permutes - full copies of the activation - do not exist."""
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.bn = nn.InstanceNorm2d(out_channels)
        self.activation = nn.LeakyReLU()
        self.linear = nn.Linear(in_channels, out_channels)

    def forward(self, x):
        if _hip_act(x):
            y = conv2d_gradfix.conv2d_bias_act(x, self.linear.weight[:, :, None, None], self.linear.bias)
        else:
            y = self.linear(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return self.activation(self.bn(y))

@persistence.persistent_class
class StyleEncoderNetworkV16(nn.Module):
    """Garment-patch encoder -> style code, plus the feature pyramid of the retained image that the synthesis blocks
    merge in (networks.py:4836-4883).  ``feat_levels`` (own extension, default = the reference's 4) is the depth of that
    pyramid: one full-resolution 3x3 layer and ``feat_levels - 1`` stride-2 layers."""
    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=4, feat_levels=4):
        super().__init__()
        plan = [(1, 2, 2), (2, 4, 2), (4, 8, 2), (8, 8, 1), (8, 8, 1), (8, 8, 1)]           # (in, out, stride) after the stem
        trunk = [Conv2dLayer(input_nc, ngf, kernel_size=1)]
        for a, b, stride in plan:
            trunk += [Dense(ngf * a, ngf * a), Conv2dLayer(ngf * a, ngf * b, kernel_size=3, down=stride)]
        self.model = nn.Sequential(*trunk, nn.AdaptiveAvgPool2d(1))
        self.fc = FullyConnectedLayer(output_nc, output_nc)
        self.feat_enc = nn.Sequential(Conv2dLayer(3, ngf, kernel_size=3),
                                      *[Conv2dLayer(ngf, ngf, kernel_size=3, down=2) for _ in range(feat_levels - 1)])

    def forward(self, x, const_input):
        pyramid = []
        for layer in self.feat_enc:
            if pyramid and _GRAD_JOIN and const_input.dtype == torch.float32:
                # a level feeds the next layer AND the synthesis blocks' merge layers: the next layer hands it on (the level again), and the
                # merge layers' gradients ride in the backward launch of its filter instead of in addition passes
                const_input, pyramid[-1] = layer(const_input, passthrough=True)
            else:
                const_input = layer(const_input)
            pyramid.append(const_input)
        code = self.fc(self.model(x).flatten(1).float())           # the style code and everything downstream of it: fp32
        return code, pyramid

#----------------------------------------------------------------------------
# SPADE blocks.

@persistence.persistent_class
class Spade_Norm_Block(torch.nn.Module):
    """InstanceNorm(x) * (1 + gamma(feat)) + beta(feat), gamma / beta = conv3x3(relu(conv3x3(feat))) (networks.py:4358-4379)."""
    def __init__(self, in_channels, norm_channels):
        super().__init__()
        self.conv_mlp = Spade_Conv2dLayer(in_channels, norm_channels, kernel_size=3, bias=False)
        self.conv_mlp_act = nn.ReLU()
        self.conv_gamma = Spade_Conv2dLayer(norm_channels, norm_channels, kernel_size=3, bias=False)
        self.conv_beta = Spade_Conv2dLayer(norm_channels, norm_channels, kernel_size=3, bias=False)
        self.param_free_norm = nn.InstanceNorm2d(norm_channels, affine=False)

    def _twin_convs(self, actv):
        g, b = self.conv_gamma, self.conv_beta
        return (actv.dtype in _HIP_DTYPES and g.weight.shape == b.weight.shape and g.bias is None and b.bias is None
                and (g.up, g.down, g.padding, g.weight_gain) == (b.up, b.down, b.padding, b.weight_gain) and g.up == g.down == 1)

    def forward(self, x, denorm_feats, post_act=None, gb=None, passthrough=False):
        relu_gain, clamp = post_act if post_act is not None else (None, None)      # the consuming layer's activation, same pass
        if gb is not None:          # (gamma | beta slice, shared gradient holder, group index) from the block's batched convolutions
            return spade_modulate(x, gb[0], None, eps=self.param_free_norm.eps, relu_gain=relu_gain, clamp=clamp, shared=(gb[1], gb[2]),
                                  passthrough=passthrough)
        mlp = self.conv_mlp         # no activation in front, nn.ReLU behind (:4373-4374): the ReLU rides in the epilogue
        actv = conv2d_resample.conv2d_resample_bias_act(x=denorm_feats, b=None, act='relu', gain=1, **mlp._resample_args(denorm_feats))
        if self._twin_convs(actv):
            # conv_gamma and conv_beta read the same tensor (:4375-4376): ONE convolution over the concatenated weights
            # writes gamma | beta as channel halves, which the normalisation kernel reads (and, backwards, writes) in place
            g = self.conv_gamma
            gamma = conv2d_resample.conv2d_resample(x=actv, w=torch.cat([g.weight, self.conv_beta.weight], dim=0), f=g.resample_filter,
                                                    padding=g.padding, flip_weight=True, wgain=g.weight_gain)
            beta = None
        else:
            gamma, beta = self.conv_gamma(actv, no_act=True), self.conv_beta(actv, no_act=True)
        return spade_modulate(x, gamma, beta, eps=self.param_free_norm.eps, relu_gain=relu_gain, clamp=clamp, passthrough=passthrough)

@persistence.persistent_class
class Spade_ResBlockV2(torch.nn.Module):
    """conv -> sqrt(1/2) * [skip(spade_skip(.)) + conv1(spade1(conv0(spade0(.))))] with activation-first layers
    (networks.py:5229-5273).  ``feat_channels`` (own extension) overrides the reference's rule for the width of the
    SPADE feature map (256 at ``resolution == 128``, else 128)."""
    def __init__(self, in_channels, out_channels, kernel_size=3, bias=True, activation='linear', up=1, down=1,
                 resample_filter=[1,3,3,1], conv_clamp=None, channels_last=False, trainable=True, resolution=128, feat_channels=None):
        super().__init__()
        _attach_filter(self, resample_filter)
        def layer(cin, cout, k):
            return Spade_Conv2dLayer(cin, cout, kernel_size=k, bias=False, resample_filter=resample_filter, conv_clamp=conv_clamp,
                                     channels_last=channels_last)
        self.conv = layer(in_channels, in_channels, 3)
        self.conv0 = layer(in_channels, out_channels, 3)
        self.conv1 = layer(out_channels, out_channels, 3)
        self.skip = layer(in_channels, out_channels, 1)
        if feat_channels is None:
            feat_channels = 256 if resolution == 128 else 128
        self.spade_skip = Spade_Norm_Block(feat_channels, in_channels)
        self.spade0 = Spade_Norm_Block(feat_channels, in_channels)
        self.spade1 = Spade_Norm_Block(feat_channels, out_channels)

    @staticmethod
    def _norm_then_conv(norm, conv, x, feat, gain, residual=None, gb=None, passthrough=False):
        """conv(norm(x, feat), gain) [+ residual]; the activation in front of the convolution is applied by the SPADE kernel when
        the layer allows it (bias-free relu).  ``passthrough``: returns ``(result, x')`` with ``x'`` = ``x`` for the block's other
        normalisation of the same tensor (its gradient joins this one's in the backward kernel: ``_SpadeModulate``)."""
        post = conv.fusable_activation(gain)
        h = norm(x, feat, gb=gb, passthrough=passthrough) if post is None else norm(x, feat, post_act=post, gb=gb, passthrough=passthrough)
        again = None
        if passthrough:
            h, again = h
        y = conv(h, gain=gain, residual=residual) if post is None else conv(h, no_act=True, residual=residual)
        return (y, again) if passthrough else y

    def _batched_gamma_beta(self, feat):
        """gamma | beta of the block's THREE normalisations from two launches (round 4): their ``conv_mlp`` layers read the same
        feature map (networks.py:4373), so one convolution over the concatenated weights computes the three hidden maps, and one GROUPED
        convolution (three groups) over the six concatenated gamma / beta weights the three gamma | beta pairs.  Backwards that is ONE input
        gradient into the feature map per block instead of three that autograd then adds (two 268 MB additions per block), and the three
        normalisations write their dgamma | dbeta as slices of one tensor (``_SharedGrad``).  None where the layers do not line up."""
        norms = (self.spade_skip, self.spade0, self.spade1)
        ok = (_SPADE_BATCH and feat.dtype == torch.float32 and feat.device.type == 'cuda' and all(n._twin_convs(feat) for n in norms)
              and len({tuple(n.conv_mlp.weight.shape) for n in norms}) == 1 and len({tuple(n.conv_gamma.weight.shape) for n in norms}) == 1
              and all(n.conv_mlp.up == n.conv_mlp.down == 1 and n.conv_mlp.bias is None for n in norms)
              and len({(n.conv_mlp.weight_gain, n.conv_gamma.weight_gain, n.conv_mlp.padding, n.conv_gamma.padding) for n in norms}) == 1)
        if not ok:
            return None, feat
        mlp, g = norms[0].conv_mlp, norms[0].conv_gamma
        conv2d_gradfix.share_pieces(feat)                # the blocks of a generator pass read ONE feature map: packed once, copied by every conv_mlp launch
        # (the feature map again as second result: the next SPADE block reads IT, and its gradient joins this convolution's input gradient
        # in that launch's epilogue -- the three blocks' gradients into the map without the two 268 MB additions autograd would make)
        actv = conv2d_resample.conv2d_resample_bias_act(x=feat, w=torch.cat([n.conv_mlp.weight for n in norms], dim=0), b=None, act='relu', gain=1,
                                                        f=mlp.resample_filter, padding=mlp.padding, flip_weight=True, wgain=mlp.weight_gain,
                                                        passthrough=_GRAD_JOIN)
        if _GRAD_JOIN:
            actv, feat = actv
        w_gb = torch.cat([w for n in norms for w in (n.conv_gamma.weight, n.conv_beta.weight)], dim=0)
        gb_all = conv2d_gradfix.conv2d(actv, w_gb, padding=g.padding, groups=3, wgain=g.weight_gain)
        holder = _SharedGrad(gb_all, 3)
        return [(v, holder, i) for i, v in enumerate(_SplitGroups.apply(gb_all, holder))], feat

    def forward(self, x, denorm_feat, return_feat=False):
        """``return_feat=True`` (own extension): returns ``(y, feat')`` -- ``feat'`` is the feature map for the NEXT block that reads it (see
        ``_batched_gamma_beta``)."""
        y = self._forward(x, denorm_feat)
        return y if return_feat else y[0]

    def _forward(self, x, denorm_feat):
        half = np.sqrt(0.5)
        x = self.conv(x, no_act=True)
        gb, feat_next = self._batched_gamma_beta(denorm_feat)
        gb = gb or (None, None, None)
        # x feeds two normalisations: the first hands it on (x again), so that the second one's input gradient is added by the first one's
        # backward kernel on its way out instead of by an addition pass over two 134 MB tensors (and the sum arrives with its maxima)
        h0, x = self._norm_then_conv(self.spade0, self.conv0, x, denorm_feat, 1, gb=gb[1], passthrough=True) if (_GRAD_JOIN and x.dtype == torch.float32) else \
            (self._norm_then_conv(self.spade0, self.conv0, x, denorm_feat, 1, gb=gb[1]), x)
        shortcut = self._norm_then_conv(self.spade_skip, self.skip, x, denorm_feat, half, gb=gb[0])
        x = h0
        if x.dtype == torch.float32 and x.device.type == 'cuda' and self.conv1.up == 1 and self.conv1.down == 1:
            # shortcut + conv1(.): the sum is formed in conv1's epilogue (the layers are activation-FIRST: the convolution is the last step).
            # fp32 storage only: in 16-bit storage the reference rounds conv1's output to the storage type BEFORE the addition, and the
            # config-5 fixtures (tests/test_config5_gpu.py) hold this path to that sequence of roundings
            return self._norm_then_conv(self.spade1, self.conv1, x, denorm_feat, half, residual=shortcut, gb=gb[2]), feat_next
        x = self._norm_then_conv(self.spade1, self.conv1, x, denorm_feat, half, gb=gb[2])
        return shortcut.add_(x), feat_next

_SPADE_BATCH = _os.environ.get('PASTA_SPADE_BATCH', '1') != '0'         # A/B switch: 0 = three conv_mlp and three gamma | beta convolutions per SPADE residual block
_MERGE_FUSED = _os.environ.get('PASTA_MERGE_FUSED', '1') != '0'         # A/B switch: 0 = torch.cat + one 1x1 convolution, as the reference

def _merge_without_cat(layer, x, side):
    """``layer(torch.cat([x, side], 1))`` for a 1x1 ``Conv2dLayer`` (networks.py:5698-5700) without the concatenated tensor: the
    pointwise convolution kernel walks its K loop over the channels of ``x`` and then of ``side`` (conv2d_gradfix.conv2d_cat1x1_bias_act).
    Against the concatenation: no 2 x 537 MB pass at 256 x 256 per call, no scan of the concatenated tensor for the operand scale, and
    the two input gradients come back as two contiguous tensors instead of channel slices of one.  (Measured and dropped, round 4,
    profiles/r4_ab_merge_split.txt: the same through linearity with the existing kernels -- conv(side, w[:, C:]) as the residual of
    conv(x, w[:, :C]) -- was 0.4 ms per step SLOWER than the concatenation: two 64-row one-tap launches cost more than the copy.)"""
    if (_MERGE_FUSED and layer.up == 1 and layer.down == 1 and layer.activation in conv2d_gradfix.FUSABLE_ACTS
            and conv2d_gradfix.cat1x1_available(x, side, layer.weight)):
        act_gain, act_clamp = _scaled_act(layer.activation, 1, layer.conv_clamp)
        return conv2d_gradfix.conv2d_cat1x1_bias_act(x, side, _master_weight(layer.weight, x), None if layer.bias is None else layer.bias.to(x.dtype),
                                                     act=layer.activation, gain=act_gain, clamp=act_clamp, wgain=layer.weight_gain)
    return layer(torch.cat([x, side], dim=1))

#----------------------------------------------------------------------------
# Full-body generator: pose-seeded style pyramid, parsing-routed SPADE stage, texture block.

class _PoseStyleBlock(torch.nn.Module):
    """One resolution of the pyramid (networks.py:5614-5719, 5313-5418).  The 4x4 block starts from the pose feature
    (its ``const`` parameter exists for state-dict compatibility and is never read); the others run an upsampling and a
    plain styled 3x3 layer; above 16x16 the retained-image feature of the block's resolution is concatenated and merged by
    a 1x1 layer; in the 'skip' architecture every block adds its ToRGB output to the upsampled running image."""
    torgb_class = None          # set by the concrete classes
    extra_outputs = 0           # outputs of torgb_class besides the image

    def _build(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture, resample_filter,
               conv_clamp, use_fp16, fp16_channels_last, torgb_kwargs, layer_kwargs):
        assert architecture in ['orig', 'skip', 'resnet']
        self.in_channels, self.w_dim, self.resolution, self.img_channels = in_channels, w_dim, resolution, img_channels
        self.is_last, self.architecture, self.use_fp16 = is_last, architecture, use_fp16
        self.channels_last = bool(use_fp16 and fp16_channels_last)
        self.half_dtype = torch.float16         # storage type of a use_fp16 block (the synthesis network may set bfloat16)
        _attach_filter(self, resample_filter)
        styled = dict(w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp, channels_last=self.channels_last, **layer_kwargs)
        first = (in_channels == 0)
        if first:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, up=2, resample_filter=resample_filter, **styled)
        self.conv1 = SynthesisLayer(out_channels, out_channels, **styled)
        self.num_conv = 1 if first else 2
        self.num_torgb = int(is_last or architecture == 'skip')
        if self.num_torgb:
            self.torgb = self.torgb_class(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp, channels_last=self.channels_last,
                                          is_last=is_last, **torgb_kwargs)
        if not first and architecture == 'resnet':
            self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=2, resample_filter=resample_filter,
                                    channels_last=self.channels_last)
        if resolution > 16:
            self.merge_conv = Conv2dLayer(out_channels + 64, out_channels, kernel_size=1, resample_filter=resample_filter,
                                          channels_last=self.channels_last)

    def forward(self, x, img, ws, pose_feature, cat_feat, force_fp32=False, fused_modconv=None, **layer_kwargs):
        misc.assert_shape(ws, [None, self.num_conv + self.num_torgb, self.w_dim])
        latents = list(ws.unbind(dim=1))
        dtype, memory_format = _block_dtype(self.use_fp16, self.channels_last, force_fp32, getattr(self, 'half_dtype', torch.float16))
        if fused_modconv is None:       # per-sample weights only outside training, and in fp16 only for a single sample
            fused_modconv = (not self.training) and (dtype == torch.float32 or int(ws.shape[0]) == 1)
        styled = dict(fused_modconv=fused_modconv, **layer_kwargs)

        if self.in_channels == 0:
            x = self.conv1(pose_feature.to(dtype=dtype, memory_format=memory_format), latents.pop(0), **styled)
        else:
            misc.assert_shape(x, [None, self.in_channels, self.resolution // 2, self.resolution // 2])
            x = x.to(dtype=dtype, memory_format=memory_format)
            if self.architecture == 'resnet':
                half = np.sqrt(0.5)
                shortcut = self.skip(x, gain=half)
                x = self.conv1(self.conv0(x, latents.pop(0), **styled), latents.pop(0), gain=half, **styled)
                x = shortcut.add_(x)
            else:
                x = self.conv1(self.conv0(x, latents.pop(0), **styled), latents.pop(0), **styled)
                if x.shape[2] > 16:
                    side = cat_feat[str(x.shape[2])].to(dtype=dtype, memory_format=memory_format)
                    x = _merge_without_cat(self.merge_conv, x, side)

        extras = (None,) * self.extra_outputs
        if img is not None:
            misc.assert_shape(img, [None, self.img_channels, self.resolution // 2, self.resolution // 2])
            img = upfirdn2d.upsample2d(img, self.resample_filter)
        if self.num_torgb:
            rgb, *extras = self.torgb(x, latents.pop(0), fused_modconv=fused_modconv)
            rgb = rgb.to(dtype=torch.float32, memory_format=torch.contiguous_format)
            extras = [e if e is None else e.to(torch.float32) for e in extras]          # every output of the network is fp32
            img = rgb if img is None else img.add_(rgb)
        return (x, img, *extras)

@persistence.persistent_class
class SynthesisBlockFull(_PoseStyleBlock):
    """networks.py:5614-5719; returns (x, img, parsing logits or None)."""
    torgb_class, extra_outputs = ToRGBLayerFull, 1

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, is_style=False, architecture='skip',
                 resample_filter=[1,3,3,1], conv_clamp=None, use_fp16=False, fp16_channels_last=False, **layer_kwargs):
        super().__init__()
        self._build(in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture, resample_filter, conv_clamp,
                    use_fp16, fp16_channels_last, dict(is_style=is_style), layer_kwargs)

@persistence.persistent_class
class SynthesisBlockV18(_PoseStyleBlock):
    """networks.py:5313-5418; returns (x, img, upper mask or None, lower mask or None)."""
    torgb_class, extra_outputs = ToRGBLayerV18, 2

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture='skip',
                 resample_filter=[1,3,3,1], conv_clamp=None, use_fp16=False, fp16_channels_last=False, **layer_kwargs):
        super().__init__()
        self._build(in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture, resample_filter, conv_clamp,
                    use_fp16, fp16_channels_last, dict(), layer_kwargs)

class _PatchRoutedSynthesis(torch.nn.Module):
    """Style pyramid b4 .. b<R>; three SPADE residual blocks at R/2 whose modulation maps come from the warped garment
    patches, routed by the pyramid's own region prediction; and a second top block (``texture_b<R>``) that renders the
    fine-tuned image from the SPADE output (networks.py:5722-5840, 5419-5531).

    The reference hard-codes R = 256 (``res == 128``, ``128*128``, ``spade_b128_*``).  Here every one of those is
    derived from ``img_resolution``: at 256 names, shapes and arithmetic are the reference's; other powers of two (the
    512x320 model of test_512.py, whose class the reference does not ship) are this package's own generalisation."""
    block_class = None
    block_kwargs_extra = staticmethod(lambda style: dict())

    def _build(self, w_dim, img_resolution, img_channels, channel_base, channel_max, block_kwargs, act_dtype=None):
        assert img_resolution >= 8 and img_resolution & (img_resolution - 1) == 0
        self.w_dim, self.img_resolution, self.img_channels = w_dim, img_resolution, img_channels
        # Own extension (BASELINE config 5): ``act_dtype`` = 'bfloat16' / 'float16' stores every activation of the synthesis
        # network in that type (convolutions: one matrix-core product, fp32 accumulation; demodulation coefficients, styles,
        # instance-norm statistics, noise strength, bias and the output images stay fp32).  None = the reference's behaviour:
        # every block computes in fp32 whatever num_fp16_res says (networks.py:5747-5748).
        self.act_dtype = _as_dtype(act_dtype)
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** k for k in range(2, self.img_resolution_log2 + 1)]
        width = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        top, below = self.block_resolutions[-1], self.block_resolutions[-2]
        self.spade_resolution = below

        def block(res, style):          # every block of the generator computes in fp32 (networks.py:5747-5748) unless act_dtype is set
            b = self.block_class(width[res // 2] if res > 4 else 0, width[res], w_dim=w_dim, resolution=res, img_channels=img_channels,
                                 is_last=(res == top), use_fp16=(self.act_dtype is not None), **self.block_kwargs_extra(style), **block_kwargs)
            if self.act_dtype is not None:
                b.half_dtype = self.act_dtype
            return b
        self.num_ws = 0
        for res in self.block_resolutions:
            b = block(res, True)
            setattr(self, f'b{res}', b)
            self.num_ws += b.num_conv + (b.num_torgb if res == top else 0)

        ngf = 64
        feat_width = 2 * (2 * ngf)                                  # upper | lower garment features of the SPADE encoder
        extra = dict() if below == 128 else dict(resolution=below, feat_channels=feat_width)
        for i in (1, 2, 3):
            setattr(self, f'spade_b{below}_{i}', Spade_ResBlockV2(width[below], width[below], **extra))
        setattr(self, f'texture_b{top}', block(top, False))
        self.spade_encoder = nn.Sequential(Conv2dLayer(3, ngf, kernel_size=7, activation='relu'),
                                           ResBlock(ngf, ngf, kernel_size=4, activation='relu'),
                                           ResBlock(ngf, 2 * ngf, kernel_size=4, activation='relu', down=2))

    def _spade_feat_parts(self, mask_256, denorm_mask, denorm_input):
        """-> (encoder features, valid mask, hole mask, divisor of the mean) of ``get_spade_feat``."""
        dt = mask_256.dtype
        halve = lambda m: torch.nn.functional.interpolate(m, scale_factor=0.5)
        region = (mask_256 > 0.9).to(dt)
        region_s = (halve(region) > 0.9).to(dt)
        covered_s = (halve(denorm_mask) > 0.9).to(dt)
        valid = ((region_s + covered_s) == 2.0).to(dt)
        hole = region_s - valid
        act = getattr(self, 'act_dtype', None) or dt
        feat = self.spade_encoder((denorm_input * region - (1 - region)).to(act))
        count = valid.sum(dim=(2, 3), keepdim=True)
        enough = (count > 10).to(dt)
        count = count * enough + float(self.spade_resolution ** 2) * (1 - enough)
        return feat, valid, hole, count

    def get_spade_feat(self, mask_256, denorm_mask, denorm_input):
        """Garment features at the SPADE resolution (networks.py:5777-5800).  ``mask_256``: the region the pyramid predicts
        for this garment, at image resolution; ``denorm_mask`` / ``denorm_input``: the warped patches and their coverage.
        Where the predicted region is not covered by a patch the feature is replaced by the mean feature of the covered
        part (of the whole map when fewer than 11 pixels are covered: the mean over ``spade_resolution^2`` positions)."""
        return self._spade_fill(*self._spade_feat_parts(mask_256, denorm_mask, denorm_input))

    @staticmethod
    def _spade_fill(feat, valid, hole, count):
        total = (feat.float() * valid).sum(dim=(2, 3), keepdim=True)
        if feat.dtype == valid.dtype:
            return feat * (1 - hole) + (total / count) * hole
        return (feat * (1 - hole).to(feat.dtype) + ((total / count) * hole).to(feat.dtype))

    def _regions(self, heads):
        """(upper, lower) garment regions at image resolution from the last block's extra ToRGB outputs."""
        raise NotImplementedError

    def _pyramid(self, ws, pose_feat, cat_feat, block_kwargs):
        misc.assert_shape(ws, [None, self.num_ws, self.w_dim])
        ws = ws.to(torch.float32)
        x = img = None
        heads, keep, start = (), None, 0
        for res in self.block_resolutions:
            b = getattr(self, f'b{res}')
            rows = ws.narrow(1, start, b.num_conv + b.num_torgb)      # a block's ToRGB shares the next block's first latent
            start += b.num_conv
            x, img, *heads = b(x, img, rows, pose_feat, cat_feat, force_fp32=(self.act_dtype is None), **block_kwargs)
            if res == self.spade_resolution:
                keep = (x.clone(), img.clone())                         # later blocks update both in place
        return img, heads, keep, rows

    def _finetune(self, keep, top_rows, heads, pose_feat, cat_feat, denorm, block_kwargs):
        du_in, dl_in, du_mask, dl_mask = denorm
        upper, lower = self._regions(heads)
        pu = self._spade_feat_parts(upper.detach(), du_mask, du_in)
        pl = self._spade_feat_parts(lower.detach(), dl_mask, dl_in)
        if _GARMENT_FUSED and pu[0].dtype == torch.float32 and pu[0].device.type == 'cuda' and pu[1].dtype == torch.float32 and pu[0].shape == pl[0].shape:
            # both garments' masked-mean fills written straight into the halves of the concatenated map: two launches instead of
            # twelve element-wise / reduction passes and a torch.cat (fp32 storage; 16-bit storage keeps the reference's roundings)
            feat = _GarmentFeat.apply(*pu, *pl)
        else:
            feat = torch.cat([self._spade_fill(*pu), self._spade_fill(*pl)], dim=1)
        x, img_below = keep
        for i in (1, 2, 3):
            x, feat = getattr(self, f'spade_b{self.spade_resolution}_{i}')(x, feat, return_feat=True)
        texture = getattr(self, f'texture_b{self.img_resolution}')
        return texture(x, img_below, top_rows, pose_feat, cat_feat, force_fp32=(self.act_dtype is None), **block_kwargs)[1]

@persistence.persistent_class
class SynthesisNetworkFull(_PatchRoutedSynthesis):
    """Training-time synthesis network (networks.py:5722-5840): regions = argmax of the 6-class parsing logits
    (class 1 upper garment, class 2 lower garment).  Returns (img, finetune_img, pred_parsing)."""
    block_class = SynthesisBlockFull
    block_kwargs_extra = staticmethod(lambda style: dict(is_style=style))

    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=0, act_dtype=None, **block_kwargs):
        super().__init__()
        self._build(w_dim, img_resolution, img_channels, channel_base, channel_max, block_kwargs, act_dtype)

    def _regions(self, heads):
        label = heads[0].detach().argmax(dim=1, keepdim=True)           # softmax is monotone: argmax of the logits (:5826)
        return (label == 1).float(), (label == 2).float()

    def forward(self, ws, pose_feat, cat_feat, denorm_upper_input, denorm_lower_input, denorm_upper_mask, denorm_lower_mask, **block_kwargs):
        img, heads, keep, top_rows = self._pyramid(ws, pose_feat, cat_feat, block_kwargs)
        denorm = (denorm_upper_input, denorm_lower_input, denorm_upper_mask, denorm_lower_mask)
        return img, self._finetune(keep, top_rows, heads, pose_feat, cat_feat, denorm, block_kwargs), heads[0]

@persistence.persistent_class
class SynthesisNetworkV18(_PatchRoutedSynthesis):
    """Synthesis network of the released inference model (networks.py:5419-5531): regions = the two sigmoid mask heads.
    Returns (img, finetune_img, upper_mask, lower_mask)."""
    block_class = SynthesisBlockV18

    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=0, act_dtype=None, **block_kwargs):
        super().__init__()
        self._build(w_dim, img_resolution, img_channels, channel_base, channel_max, block_kwargs, act_dtype)

    def _regions(self, heads):
        return heads[0], heads[1]

    def forward(self, ws, pose_feat, cat_feat, denorm_upper_input, denorm_lower_input, denorm_upper_mask, denorm_lower_mask, **block_kwargs):
        img, heads, keep, top_rows = self._pyramid(ws, pose_feat, cat_feat, block_kwargs)
        denorm = (denorm_upper_input, denorm_lower_input, denorm_upper_mask, denorm_lower_mask)
        return img, self._finetune(keep, top_rows, heads, pose_feat, cat_feat, denorm, block_kwargs), heads[0], heads[1]

class _TryOnGenerator(torch.nn.Module):
    """Pose encoder + garment-patch style encoder + mapping + synthesis (networks.py:5843-5881, 5534-5577).
    At 256 the sub-networks have the reference's depths (6 pose stages -> 4x4, 4 retained-image features for the merges
    at 32..256); other resolutions scale both with log2(img_resolution) (own generalisation, see _PatchRoutedSynthesis)."""
    synthesis_class = None
    patch_channels = None       # channels of the stacked garment patches fed to the style encoder

    def _build(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs, synthesis_kwargs):
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.synthesis = self.synthesis_class(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)
        log2 = int(np.log2(img_resolution))
        self.const_encoding = ConstEncoderNetwork(input_nc=3 + 3, output_nc=512, ngf=64, n_downsampling=log2 - 2)
        levels = dict() if log2 == 8 else dict(feat_levels=log2 - 4)
        self.style_encoding = StyleEncoderNetworkV16(input_nc=self.patch_channels, output_nc=512, ngf=64, n_downsampling=6, **levels)

    def forward(self, z, c, retain, pose, denorm_upper_input, denorm_lower_input, denorm_upper_mask, denorm_lower_mask,
                truncation_psi=1, truncation_cutoff=None, **synthesis_kwargs):
        act = getattr(self.synthesis, 'act_dtype', None)
        if act is not None:             # 16-bit activation storage (own extension): the encoders run in it as well
            pose, c, retain = pose.to(act), c.to(act), retain.to(act)
        pose_feat = self.const_encoding(pose)
        code, pyramid = self.style_encoding(c, retain)
        ws = self.mapping(z, code, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        by_size = {str(f.shape[2]): f for f in pyramid}
        return self.synthesis(ws, pose_feat, by_size, denorm_upper_input, denorm_lower_input, denorm_upper_mask, denorm_lower_mask,
                              **synthesis_kwargs)

@persistence.persistent_class
class GeneratorFull(_TryOnGenerator):
    """The training generator (networks.py:5843-5881): 14 garment patches x 3 channels."""
    synthesis_class, patch_channels = SynthesisNetworkFull, 10 * 3 + 4 * 3

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, synthesis_kwargs={}):
        super().__init__()
        self._build(z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs, synthesis_kwargs)

@persistence.persistent_class
class GeneratorV18(_TryOnGenerator):
    """The generator inside test.py's released pickle (networks.py:5534-5577): 60-channel patch stack."""
    synthesis_class, patch_channels = SynthesisNetworkV18, 30 * 2

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, synthesis_kwargs={}):
        super().__init__()
        self._build(z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs, synthesis_kwargs)

#----------------------------------------------------------------------------
# Discriminator.

@persistence.persistent_class
class DiscriminatorBlock(torch.nn.Module):
    """[fromrgb ->] 3x3 -> 3x3 /2, with a 1x1 /2 residual branch in the 'resnet' architecture (networks.py:916-996)."""
    def __init__(self, in_channels, tmp_channels, out_channels, resolution, img_channels, first_layer_idx, architecture='resnet',
                 activation='lrelu', resample_filter=[1,3,3,1], conv_clamp=None, use_fp16=False, fp16_channels_last=False, freeze_layers=0):
        assert in_channels in [0, tmp_channels]
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels, self.resolution, self.img_channels = in_channels, resolution, img_channels
        self.first_layer_idx, self.architecture, self.use_fp16 = first_layer_idx, architecture, use_fp16
        self.channels_last = bool(use_fp16 and fp16_channels_last)
        self.half_dtype = torch.float16         # storage type of a use_fp16 block (Discriminator(half_dtype=...) may set bfloat16)
        _attach_filter(self, resample_filter)
        # Freeze-D: layers are numbered through the whole discriminator; those below ``freeze_layers`` hold buffers
        plan = []
        if in_channels == 0 or architecture == 'skip':
            plan.append(('fromrgb', img_channels, tmp_channels, dict(kernel_size=1, activation=activation, conv_clamp=conv_clamp)))
        plan.append(('conv0', tmp_channels, tmp_channels, dict(kernel_size=3, activation=activation, conv_clamp=conv_clamp)))
        plan.append(('conv1', tmp_channels, out_channels, dict(kernel_size=3, activation=activation, down=2, conv_clamp=conv_clamp,
                                                               resample_filter=resample_filter)))
        if architecture == 'resnet':
            plan.append(('skip', tmp_channels, out_channels, dict(kernel_size=1, bias=False, down=2, resample_filter=resample_filter)))
        for offset, (name, cin, cout, kw) in enumerate(plan):
            setattr(self, name, Conv2dLayer(cin, cout, trainable=(first_layer_idx + offset >= freeze_layers),
                                            channels_last=self.channels_last, **kw))
        self.num_layers = len(plan)

    def forward(self, x, img, force_fp32=False):
        dtype, memory_format = _block_dtype(self.use_fp16, self.channels_last, force_fp32, getattr(self, 'half_dtype', torch.float16))
        if x is not None:
            misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
            x = x.to(dtype=dtype, memory_format=memory_format)
        if hasattr(self, 'fromrgb'):
            misc.assert_shape(img, [None, self.img_channels, self.resolution, self.resolution])
            img = img.to(dtype=dtype, memory_format=memory_format)
            y = self.fromrgb(img)
            x = y if x is None else x + y
            img = upfirdn2d.downsample2d(img, self.resample_filter) if self.architecture == 'skip' else None
        if self.architecture == 'resnet':
            half = np.sqrt(0.5)
            h, x = _layer_and_input(self.conv0, x)      # x again: the skip branch's gradient joins conv0's input gradient in that launch
            x = self.skip(x, gain=half, add=self.conv1(h, gain=half))     # skip(x) + conv1(.): the sum in the skip convolution's epilogue
        else:
            x = self.conv1(self.conv0(x))
        assert x.dtype == dtype
        return x, img

@persistence.persistent_class
class MinibatchStdLayer(torch.nn.Module):
    """Appends, per group of ``group_size`` samples, the feature standard deviation averaged over channels (in
    ``num_channels`` slices) and pixels, as constant extra channels (networks.py:1000-1022).  Groups are strided: sample m
    belongs with m + B, m + 2B, ... where B = N / group size."""
    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size, self.num_channels = group_size, num_channels

    def forward(self, x):
        n, c, h, w = x.shape
        g = int(n) if self.group_size is None else min(int(self.group_size), int(n))
        f = self.num_channels
        grouped = x.reshape(g, n // g, f, c // f, h, w)
        std = (grouped.var(dim=0, unbiased=False) + 1e-8).sqrt()                # [B, F, c/F, H, W]
        stat = std.mean(dim=[2, 3, 4]).reshape(n // g, f, 1, 1)
        return torch.cat([x, stat.repeat(g, 1, h, w)], dim=1)

@persistence.persistent_class
class DiscriminatorEpilogue(torch.nn.Module):
    """4x4 tail: minibatch-std -> 3x3 -> FC -> FC, then the projection onto the mapped conditioning vector
    (networks.py:1026-1080)."""
    def __init__(self, in_channels, cmap_dim, resolution, img_channels, architecture='resnet', mbstd_group_size=4,
                 mbstd_num_channels=1, activation='lrelu', conv_clamp=None):
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels, self.cmap_dim, self.resolution = in_channels, cmap_dim, resolution
        self.img_channels, self.architecture = img_channels, architecture
        if architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, in_channels, kernel_size=1, activation=activation)
        self.mbstd = MinibatchStdLayer(group_size=mbstd_group_size, num_channels=mbstd_num_channels) if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation, conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * resolution * resolution, in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, cmap_dim if cmap_dim > 0 else 1)

    def forward(self, x, img, cmap, force_fp32=False):
        del force_fp32                  # the tail always computes in fp32
        misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
        x = x.to(dtype=torch.float32, memory_format=torch.contiguous_format)
        if self.architecture == 'skip':
            misc.assert_shape(img, [None, self.img_channels, self.resolution, self.resolution])
            x = x + self.fromrgb(img.to(dtype=torch.float32, memory_format=torch.contiguous_format))
        if self.mbstd is not None:
            x = self.mbstd(x)
        x = self.out(self.fc(self.conv(x).flatten(1)))
        if self.cmap_dim > 0:
            misc.assert_shape(cmap, [None, self.cmap_dim])
            x = (x * cmap).sum(dim=1, keepdim=True) / np.sqrt(self.cmap_dim)
        assert x.dtype == torch.float32
        return x

@persistence.persistent_class
class Discriminator(torch.nn.Module):
    """Residual StyleGAN2 discriminator, projection-conditioned on the style code (networks.py:1084-1139).  The
    ``num_fp16_res`` highest resolutions compute in fp16 (none when 0)."""
    def __init__(self, c_dim, img_resolution, img_channels, architecture='resnet', channel_base=32768, channel_max=512,
                 num_fp16_res=0, conv_clamp=None, cmap_dim=None, block_kwargs={}, mapping_kwargs={}, epilogue_kwargs={}, half_dtype='float16'):
        super().__init__()
        self.c_dim, self.img_resolution, self.img_channels = c_dim, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** k for k in range(self.img_resolution_log2, 2, -1)]
        width = {res: min(channel_base // res, channel_max) for res in self.block_resolutions + [4]}
        first_fp16 = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)
        if c_dim == 0:
            cmap_dim = 0
        elif cmap_dim is None:
            cmap_dim = width[4]
        shared = dict(img_channels=img_channels, architecture=architecture, conv_clamp=conv_clamp)
        layer_idx = 0
        for res in self.block_resolutions:
            blk = DiscriminatorBlock(width[res] if res < img_resolution else 0, width[res], width[res // 2], resolution=res,
                                     first_layer_idx=layer_idx, use_fp16=(res >= first_fp16), **block_kwargs, **shared)
            blk.half_dtype = _as_dtype(half_dtype) or torch.float16       # own extension: 'bfloat16' for BASELINE config 5
            setattr(self, f'b{res}', blk)
            layer_idx += blk.num_layers
        if c_dim > 0:
            self.mapping = MappingNetwork(z_dim=0, c_dim=c_dim, w_dim=cmap_dim, num_ws=None, w_avg_beta=None, **mapping_kwargs)
        self.b4 = DiscriminatorEpilogue(width[4], cmap_dim=cmap_dim, resolution=4, **epilogue_kwargs, **shared)

    def forward(self, img, c, **block_kwargs):
        x = None
        for res in self.block_resolutions:
            x, img = getattr(self, f'b{res}')(x, img, **block_kwargs)
        return self.b4(x, img, self.mapping(None, c) if self.c_dim > 0 else None)

#----------------------------------------------------------------------------
