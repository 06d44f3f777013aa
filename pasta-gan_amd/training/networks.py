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

import numpy as np
import torch
import torch.nn as nn

from torch_utils import misc
from torch_utils import persistence
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
            ds = fma.plane_dot(dy, x).reshape(s.shape)
        return dx, ds

def scale_planes(x, s):
    """x * s.reshape(N, C, 1, 1) in one pass (fp32 NCHW on the GPU); other dtypes use a broadcast multiply."""
    if x.dtype == torch.float32 and x.device.type == 'cuda' and x.ndim == 4 and x.numel() > 0:
        return _ScalePlanes.apply(x, s.to(torch.float32).reshape(x.shape[0], x.shape[1]))
    return x * s.to(x.dtype).reshape(x.shape[0], -1, 1, 1)

class _SpadeModulate(torch.autograd.Function):
    """InstanceNorm(x) * (1 + gamma) + beta with statistics, normalisation and modulation in one
    kernel (networks.py:4371-4379: InstanceNorm2d(affine=False), eps 1e-5, biased variance).
    ``beta is None``: ``gamma`` is a [N, 2C, H, W] tensor holding gamma in its first C channels and beta in its last C (the
    output of ONE convolution with the concatenated conv_gamma / conv_beta weights); its gradient comes back as one tensor,
    so the two input gradients of that convolution accumulate inside its K loop instead of in an addition pass."""
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, post):
        n, c, h, w = x.shape
        x, gamma = x.contiguous(), gamma.contiguous()
        fused = beta is None
        if fused:
            assert gamma.shape == (n, 2 * c, h, w)
            beta_ptr, gstride = gamma.data_ptr() + 4 * c * h * w, 2 * c * h * w
        else:
            beta = beta.contiguous()
            beta_ptr, gstride = beta.data_ptr(), 0
        out = torch.empty_like(x)
        stats = torch.empty([n * c, 2], dtype=torch.float32, device=x.device)
        act, gain, clamp = post                     # (2, gain, clamp): relu * gain with clamp on the way out; (0, 1, -1): none
        with torch.cuda.device(x.device):
            st = _native.lib().pasta_spade_norm(_native.ptr(x), _native.ptr(gamma), beta_ptr, _native.ptr(out),
                                                _native.ptr(stats), n * c, h * w, float(eps), act, float(gain), float(clamp),
                                                c, gstride, _native.stream())
        _native.check(st)
        ctx.save_for_backward(x, gamma, stats, beta if (act == 2 and not fused) else None)
        ctx.post, ctx.fused = post, fused
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        x, gamma, stats, beta = ctx.saved_tensors
        act, gain, clamp = ctx.post
        n, c, h, w = x.shape
        dout = dout.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        lib = _native.lib()
        if ctx.fused:       # gamma | beta and their gradients as channel halves of one tensor each
            half = 4 * c * h * w
            dgb = torch.empty_like(gamma) if ctx.needs_input_grad[1] else None
            if dx is not None or dgb is not None:
                with torch.cuda.device(x.device):
                    st = lib.pasta_spade_norm_bwd(_native.ptr(dout), _native.ptr(x), _native.ptr(gamma), _native.ptr(stats), _native.ptr(dx),
                                                  _native.ptr(dgb), dgb.data_ptr() + half if dgb is not None else None, n * c, h * w,
                                                  gamma.data_ptr() + half, act, float(gain), float(clamp), c, 2 * c * h * w, 2 * c * h * w,
                                                  _native.stream())
                _native.check(st)
            return dx, dgb, None, None, None
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
                                              n * c, h * w, _native.ptr(beta), act, float(gain), float(clamp), c, 0, 0, _native.stream())
            _native.check(st)
        return dx, dgamma, (dbeta if ctx.needs_input_grad[2] else None), None, None

def spade_modulate(x, gamma, beta, eps=1e-5, relu_gain=None, clamp=None):
    """InstanceNorm(x) * (1 + gamma) + beta; ``relu_gain`` not None additionally applies ``min(relu(.) * relu_gain, clamp)``
    in the same pass (the activation of the Spade_Conv2dLayer that consumes the result).  ``beta=None``: ``gamma`` holds
    gamma | beta as the two channel halves of a [N, 2C, H, W] tensor."""
    _native.require_gpu(x, 'spade_modulate')
    if x.dtype != torch.float32:
        raise RuntimeError('spade_modulate: float32 only (the generator runs in fp32, networks.py:5747-5748)')
    post = (0, 1.0, -1.0) if relu_gain is None else (2, float(relu_gain), float(clamp if clamp is not None else -1))
    return _SpadeModulate.apply(x, gamma, beta, eps, post)

#----------------------------------------------------------------------------

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
        with torch.cuda.device(u.device):
            st = _native.lib().pasta_mod_bias_act(_native.ptr(u), _native.ptr(d), _native.ptr(noise), _native.ptr(strength), _native.ptr(b),
                                                  _native.ptr(y), n, c, h * w, per_sample, act_idx, float(alpha), float(gain), float(clamp),
                                                  _native.stream())
        _native.check(st)
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
        part = torch.empty([lib.pasta_mod_bias_act_bwd_workspace(n, c, h * w) // 4], dtype=torch.float32, device=u.device)
        with torch.cuda.device(u.device):
            st = lib.pasta_mod_bias_act_bwd(_native.ptr(dy), _native.ptr(y), _native.ptr(u), _native.ptr(d), _native.ptr(noise), _native.ptr(du),
                                            _native.ptr(part), n, c, h * w, ctx.per_sample, act_idx, float(alpha), float(gain), float(clamp),
                                            _native.stream())
        _native.check(st)
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
    assert act in ('linear', 'lrelu') and u.dtype == torch.float32
    cfg = (spec.cuda_idx, float(alpha if alpha is not None else spec.def_alpha), float(gain if gain is not None else spec.def_gain),
           float(clamp if clamp is not None else -1))
    return _ModBiasAct.apply(u, dcoefs, noise, strength, bias, cfg)

#----------------------------------------------------------------------------

@misc.profiled_function
def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """networks.py:30-32"""
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()

#----------------------------------------------------------------------------

def _modulate_and_convolve(x, weight, styles, up, down, padding, resample_filter, demodulate, flip_weight):
    """Non-fused modulated convolution up to, not including, the demodulation (networks.py:65-76): returns the
    convolution of the style-scaled activations and the demodulation coefficients ``rsqrt(s^2 @ sum_k w^2 + 1e-8)``."""
    dcoefs = None
    if demodulate:
        wsq = weight.square().sum(dim=[2, 3])                                   # [O, I]
        dcoefs = (styles.square() @ wsq.t() + 1e-8).rsqrt()                     # [N, O]
    x = scale_planes(x, styles)
    x = conv2d_resample.conv2d_resample(x=x, w=weight.to(x.dtype), f=resample_filter, up=up, down=down,
                                        padding=padding, flip_weight=flip_weight)
    return x, dcoefs

@misc.profiled_function
def modulated_conv2d(
    x,                          # Input tensor of shape [batch_size, in_channels, in_height, in_width].
    weight,                     # Weight tensor of shape [out_channels, in_channels, kernel_height, kernel_width].
    styles,                     # Modulation coefficients of shape [batch_size, in_channels].
    noise           = None,     # Optional noise tensor to add to the output activations.
    up              = 1,        # Integer upsampling factor.
    down            = 1,        # Integer downsampling factor.
    padding         = 0,        # Padding with respect to the upsampled image.
    resample_filter = None,     # Low-pass filter to apply when resampling activations (upfirdn2d.setup_filter()).
    demodulate      = True,     # Apply weight demodulation?
    flip_weight     = True,     # False = convolution, True = correlation (matches torch.nn.functional.conv2d).
    fused_modconv   = True,     # Perform modulation, convolution, and demodulation as a single fused operation?
):
    """StyleGAN2 modulated convolution (networks.py:36-94).

    ``fused_modconv=False`` (training): scale activations by the styles, run one shared-weight
    convolution, scale by the demodulation coefficients and add noise. The coefficients
    ``rsqrt(sum_{i,k} (w*s)^2 + 1e-8)`` are evaluated as ``rsqrt(s^2 @ sum_k w^2 + 1e-8)`` so the
    per-sample weight tensor [N,O,I,k,k] is never materialised.
    ``fused_modconv=True`` (inference): per-sample weights and one grouped convolution."""
    batch_size = x.shape[0]
    out_channels, in_channels, kh, kw = weight.shape
    misc.assert_shape(weight, [out_channels, in_channels, kh, kw])
    misc.assert_shape(x, [batch_size, in_channels, None, None])
    misc.assert_shape(styles, [batch_size, in_channels])

    # Pre-normalize inputs to avoid FP16 overflow (networks.py:57-59).
    if x.dtype == torch.float16 and demodulate:
        weight = weight * (1 / np.sqrt(in_channels * kh * kw) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)

    if not fused_modconv:
        x, dcoefs = _modulate_and_convolve(x, weight, styles, up, down, padding, resample_filter, demodulate, flip_weight)
        if demodulate and noise is not None:
            x = fma.fma(x, dcoefs.to(x.dtype).reshape(batch_size, -1, 1, 1), noise.to(x.dtype))
        elif demodulate:
            x = scale_planes(x, dcoefs)
        elif noise is not None:
            x = x.add_(noise.to(x.dtype))
        return x

    # One grouped convolution with per-sample weights (networks.py:84-94).
    w = weight.unsqueeze(0) * styles.reshape(batch_size, 1, -1, 1, 1)               # [N, O, I, k, k]
    if demodulate:
        dcoefs = (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
        w = w * dcoefs.reshape(batch_size, -1, 1, 1, 1)
    batch_size = int(batch_size)
    x = x.reshape(1, -1, *x.shape[2:])
    w = w.reshape(-1, in_channels, kh, kw)
    x = conv2d_resample.conv2d_resample(x=x, w=w.to(x.dtype), f=resample_filter, up=up, down=down, padding=padding,
                                        groups=batch_size, flip_weight=flip_weight)
    x = x.reshape(batch_size, -1, *x.shape[2:])
    if noise is not None:
        x = x.add_(noise)
    return x

#----------------------------------------------------------------------------

@persistence.persistent_class
class FullyConnectedLayer(torch.nn.Module):
    """networks.py:98-128"""
    def __init__(self,
        in_features,                # Number of input features.
        out_features,               # Number of output features.
        bias            = True,     # Apply additive bias before the activation function?
        activation      = 'linear', # Activation function: 'relu', 'lrelu', etc.
        lr_multiplier   = 1,        # Learning rate multiplier.
        bias_init       = 0,        # Initial value for the additive bias.
    ):
        super().__init__()
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        # networks.py:115-128.  The weight and bias gains ride in the GEMM's alpha / beta instead of two scaling kernels.
        w = self.weight.to(x.dtype)
        b = self.bias
        if b is not None:
            y = torch.addmm(b.to(x.dtype).unsqueeze(0), x, w.t(), beta=float(self.bias_gain), alpha=float(self.weight_gain))
            return y if self.activation == 'linear' else bias_act.bias_act(y, None, act=self.activation)
        x = x.matmul((w * self.weight_gain).t())
        return bias_act.bias_act(x, None, act=self.activation)

#----------------------------------------------------------------------------

def _make_conv_params(module, in_channels, out_channels, kernel_size, bias, channels_last, trainable):
    """weight / bias as Parameters (trainable) or buffers (frozen), networks.py:158-168."""
    memory_format = torch.channels_last if channels_last else torch.contiguous_format
    weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format)
    bias = torch.zeros([out_channels]) if bias else None
    if trainable:
        module.weight = torch.nn.Parameter(weight)
        module.bias = torch.nn.Parameter(bias) if bias is not None else None
    else:
        module.register_buffer('weight', weight)
        if bias is not None:
            module.register_buffer('bias', bias)
        else:
            module.bias = None

@persistence.persistent_class
class Conv2dLayer(torch.nn.Module):
    """conv2d_resample -> bias_act (networks.py:132-179)."""
    def __init__(self,
        in_channels,                    # Number of input channels.
        out_channels,                   # Number of output channels.
        kernel_size,                    # Width and height of the convolution kernel.
        bias            = True,         # Apply additive bias before the activation function?
        activation      = 'linear',     # Activation function: 'relu', 'lrelu', etc.
        up              = 1,            # Integer upsampling factor.
        down            = 1,            # Integer downsampling factor.
        resample_filter = [1,3,3,1],    # Low-pass filter to apply when resampling activations.
        conv_clamp      = None,         # Clamp the output to +-X, None = disable clamping.
        channels_last   = False,        # Expect the input to have memory_format=channels_last?
        trainable       = True,         # Update the weights of this layer during training?
    ):
        super().__init__()
        self.activation = activation
        self.up = up
        self.down = down
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        _make_conv_params(self, in_channels, out_channels, kernel_size, bias, channels_last, trainable)

    def forward(self, x, gain=1):
        # `w = self.weight * self.weight_gain` (networks.py:171) is folded into the convolution's weight packing (wgain)
        b = self.bias.to(x.dtype) if self.bias is not None else None
        flip_weight = (self.up == 1)
        act_gain = self.act_gain * gain
        act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return conv2d_resample.conv2d_resample_bias_act(x=x, w=self.weight.to(x.dtype), b=b, f=self.resample_filter, up=self.up,
                                                        down=self.down, padding=self.padding, flip_weight=flip_weight,
                                                        act=self.activation, gain=act_gain, clamp=act_clamp, wgain=self.weight_gain)

#----------------------------------------------------------------------------

@persistence.persistent_class
class MappingNetwork(torch.nn.Module):
    """Label embedding + FC stack -> w, with the running average ``w_avg`` (networks.py:183-259)."""
    def __init__(self,
        z_dim,                      # Input latent (Z) dimensionality, 0 = no latent.
        c_dim,                      # Conditioning label (C) dimensionality, 0 = no label.
        w_dim,                      # Intermediate latent (W) dimensionality.
        num_ws,                     # Number of intermediate latents to output, None = do not broadcast.
        num_layers      = 8,        # Number of mapping layers.
        embed_features  = None,     # Label embedding dimensionality, None = same as w_dim.
        layer_features  = None,     # Number of intermediate features in the mapping layers, None = same as w_dim.
        activation      = 'lrelu',  # Activation function: 'relu', 'lrelu', etc.
        lr_multiplier   = 0.01,     # Learning rate multiplier for the mapping layers.
        w_avg_beta      = 0.995,    # Decay for tracking the moving average of W during training, None = do not track.
    ):
        super().__init__()
        self.z_dim = z_dim
        self.c_dim = c_dim
        self.w_dim = w_dim
        self.num_ws = num_ws
        self.num_layers = num_layers
        self.w_avg_beta = w_avg_beta
        if embed_features is None:
            embed_features = w_dim
        if c_dim == 0:
            embed_features = 0
        if layer_features is None:
            layer_features = w_dim
        features = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', FullyConnectedLayer(features[idx], features[idx + 1], activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False):
        x = None
        if self.z_dim > 0:
            misc.assert_shape(z, [None, self.z_dim])
            x = normalize_2nd_moment(z.to(torch.float32))
        if self.c_dim > 0:
            misc.assert_shape(c, [None, self.c_dim])
            y = normalize_2nd_moment(self.embed(c.to(torch.float32)))
            x = torch.cat([x, y], dim=1) if x is not None else y
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x)
        if self.w_avg_beta is not None and self.training and not skip_w_avg_update:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            assert self.w_avg_beta is not None
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x

#----------------------------------------------------------------------------

@persistence.persistent_class
class SynthesisLayer(torch.nn.Module):
    """Affine -> modulated conv (+noise) -> bias_act (networks.py:263-315)."""
    def __init__(self,
        in_channels,                    # Number of input channels.
        out_channels,                   # Number of output channels.
        w_dim,                          # Intermediate latent (W) dimensionality.
        resolution,                     # Resolution of this layer.
        kernel_size     = 3,            # Convolution kernel size.
        up              = 1,            # Integer upsampling factor.
        use_noise       = True,         # Enable noise input?
        activation      = 'lrelu',      # Activation function: 'relu', 'lrelu', etc.
        resample_filter = [1,3,3,1],    # Low-pass filter to apply when resampling activations.
        conv_clamp      = None,         # Clamp the output of convolution layers to +-X, None = disable clamping.
        channels_last   = False,        # Use channels_last format for the weights?
    ):
        super().__init__()
        self.resolution = resolution
        self.up = up
        self.use_noise = use_noise
        self.activation = activation
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        memory_format = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format))
        if use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1):
        assert noise_mode in ['random', 'const', 'none']
        in_resolution = self.resolution // self.up
        misc.assert_shape(x, [None, self.weight.shape[1], in_resolution, in_resolution])
        styles = self.affine(w)
        flip_weight = (self.up == 1)
        act_gain = self.act_gain * gain
        act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        if not fused_modconv and x.dtype == torch.float32 and x.device.type == 'cuda' and self.activation in ('linear', 'lrelu'):
            # training path: demodulation, noise, bias and activation as one pass over the convolution's output
            unit = None
            if self.use_noise and noise_mode == 'random':
                unit = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device)
            if self.use_noise and noise_mode == 'const':
                unit = self.noise_const
            u, dcoefs = _modulate_and_convolve(x, self.weight, styles, self.up, 1, self.padding, self.resample_filter, True, flip_weight)
            return mod_bias_act(u, dcoefs, unit, self.noise_strength if unit is not None else None, self.bias, act=self.activation,
                                gain=act_gain, clamp=act_clamp)
        noise = None
        if self.use_noise and noise_mode == 'random':
            noise = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device) * self.noise_strength
        if self.use_noise and noise_mode == 'const':
            noise = self.noise_const * self.noise_strength
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, noise=noise, up=self.up, padding=self.padding,
                             resample_filter=self.resample_filter, flip_weight=flip_weight, fused_modconv=fused_modconv)
        return bias_act.bias_act(x, self.bias.to(x.dtype), act=self.activation, gain=act_gain, clamp=act_clamp)

#----------------------------------------------------------------------------

@persistence.persistent_class
class ToRGBLayer(torch.nn.Module):
    """1x1 modulated conv without demodulation -> bias_act(linear, clamp) (networks.py:319-334)."""
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        memory_format = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))

    def forward(self, x, w, fused_modconv=True):
        styles = self.affine(w) * self.weight_gain
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, demodulate=False, fused_modconv=fused_modconv)
        return bias_act.bias_act(x, self.bias.to(x.dtype), clamp=self.conv_clamp)

@persistence.persistent_class
class ToRGBLayerFull(torch.nn.Module):
    """ToRGB with an extra 6-class parsing head on the last style block (networks.py:5582-5611)."""
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False, is_last=False, is_style=False):
        super().__init__()
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        memory_format = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.is_last = is_last
        self.is_style = is_style
        if self.is_last and self.is_style:
            self.m_weight1 = torch.nn.Parameter(torch.randn([6, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format))
            self.m_bias1 = torch.nn.Parameter(torch.zeros([6]))

    def forward(self, x, w, fused_modconv=True):
        styles = self.affine(w) * self.weight_gain
        pred_parsing = None
        if self.is_last and self.is_style:
            pred_parsing = modulated_conv2d(x=x, weight=self.m_weight1, styles=styles, demodulate=False, fused_modconv=fused_modconv)
            pred_parsing = bias_act.bias_act(pred_parsing, self.m_bias1.to(x.dtype), clamp=self.conv_clamp)
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, demodulate=False, fused_modconv=fused_modconv)
        x = bias_act.bias_act(x, self.bias.to(x.dtype), clamp=self.conv_clamp)
        return x, pred_parsing

@persistence.persistent_class
class ToRGBLayerV18(torch.nn.Module):
    """ToRGB of the released 256 inference model: the last block also predicts sigmoid upper/lower clothing
    masks (networks.py:5276-5310)."""
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False, is_last=False):
        super().__init__()
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        memory_format = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.is_last = is_last
        if self.is_last:
            self.m_weight1 = torch.nn.Parameter(torch.randn([1, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format))
            self.m_bias1 = torch.nn.Parameter(torch.zeros([1]))
            self.m_weight2 = torch.nn.Parameter(torch.randn([1, in_channels, kernel_size, kernel_size]).to(memory_format=memory_format))
            self.m_bias2 = torch.nn.Parameter(torch.zeros([1]))

    def forward(self, x, w, fused_modconv=True):
        styles = self.affine(w) * self.weight_gain
        upper_mask = lower_mask = None
        if self.is_last:
            upper_mask = modulated_conv2d(x=x, weight=self.m_weight1, styles=styles, demodulate=False, fused_modconv=fused_modconv)
            upper_mask = bias_act.bias_act(upper_mask, self.m_bias1.to(x.dtype), clamp=self.conv_clamp, act='sigmoid')
            lower_mask = modulated_conv2d(x=x, weight=self.m_weight2, styles=styles, demodulate=False, fused_modconv=fused_modconv)
            lower_mask = bias_act.bias_act(lower_mask, self.m_bias2.to(x.dtype), clamp=self.conv_clamp, act='sigmoid')
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, demodulate=False, fused_modconv=fused_modconv)
        x = bias_act.bias_act(x, self.bias.to(x.dtype), clamp=self.conv_clamp)
        return x, upper_mask, lower_mask

#----------------------------------------------------------------------------
# Encoders.

@persistence.persistent_class
class ResBlock(torch.nn.Module):
    """3x3 -> 3x3 with a 1x1 skip, each branch scaled by sqrt(1/2) (networks.py:528-558)."""
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=[1,3,3,1], conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        common = dict(resample_filter=resample_filter, conv_clamp=conv_clamp, channels_last=channels_last)
        self.conv0 = Conv2dLayer(in_channels, out_channels, kernel_size=3, activation=activation, up=up, down=down, bias=bias, **common)
        self.conv1 = Conv2dLayer(out_channels, out_channels, kernel_size=3, activation=activation, bias=bias, **common)
        self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=up, down=down, **common)

    def forward(self, x):
        y = self.skip(x, gain=np.sqrt(0.5))
        x = self.conv0(x)
        x = self.conv1(x, gain=np.sqrt(0.5))
        return y.add_(x)

@persistence.persistent_class
class ConstEncoderNetwork(nn.Module):
    """Pose encoder: 1x1 stem then ``n_downsampling`` stride-2 3x3 convs (networks.py:560-579)."""
    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=4):
        super().__init__()
        mult_ins = [1, 2, 4, 4, 4, 8]
        mult_outs = [2, 4, 4, 4, 8, 8]
        layers = [Conv2dLayer(input_nc, ngf, kernel_size=1)]
        for i in range(n_downsampling):
            layers.append(Conv2dLayer(ngf * mult_ins[i], ngf * mult_outs[i], kernel_size=3, down=2))
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)

class Dense(nn.Module):
    """Per-pixel Linear -> InstanceNorm -> LeakyReLU(0.01) (networks.py:594-611)."""
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.bn = nn.InstanceNorm2d(out_channels)
        self.activation = nn.LeakyReLU()
        self.linear = nn.Linear(in_channels, out_channels)

    def forward(self, x):
        out = self.linear(x.permute((0, 2, 3, 1))).permute((0, 3, 1, 2))
        return self.activation(self.bn(out))

@persistence.persistent_class
class StyleEncoderNetworkV16(nn.Module):
    """Patch style encoder (-> 512-d code) plus the retain-image feature pyramid (networks.py:4836-4883)."""
    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=4):
        super().__init__()
        encoder = [Conv2dLayer(input_nc, ngf, kernel_size=1)]
        for mult_in, mult_out in zip([1, 2, 4], [2, 4, 8]):
            encoder += [Dense(ngf * mult_in, ngf * mult_in), Conv2dLayer(ngf * mult_in, ngf * mult_out, kernel_size=3, down=2)]
        for mult_in, mult_out in zip([8, 8, 8], [8, 8, 8]):
            encoder += [Dense(ngf * mult_in, ngf * mult_in), Conv2dLayer(ngf * mult_in, ngf * mult_out, kernel_size=3)]
        encoder += [nn.AdaptiveAvgPool2d(1)]
        self.model = nn.Sequential(*encoder)
        self.fc = FullyConnectedLayer(output_nc, output_nc)
        feat_enc = [Conv2dLayer(3, ngf, kernel_size=3)]
        for _ in range(3):
            feat_enc += [Conv2dLayer(ngf, ngf, kernel_size=3, down=2)]
        self.feat_enc = nn.Sequential(*feat_enc)

    def forward(self, x, const_input):
        const_feats = []
        for module in self.feat_enc:
            const_input = module(const_input)
            const_feats.append(const_input)
        for module in self.model:
            x = module(x)
        x = self.fc(x.view(x.size(0), -1))
        return x, const_feats

#----------------------------------------------------------------------------
# SPADE blocks.

@persistence.persistent_class
class Spade_Conv2dLayer(torch.nn.Module):
    """Activation-before-convolution layer (networks.py:4304-4355)."""
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='relu', up=1, down=1,
                 resample_filter=[1,3,3,1], conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        self.activation = activation
        self.up = up
        self.down = down
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        _make_conv_params(self, in_channels, out_channels, kernel_size, bias, channels_last, trainable)

    def fusable_activation(self, gain=1):
        """(relu gain, clamp) of the activation this layer applies in front of its convolution, when the producer of
        its input can apply it instead (no bias, relu); else None."""
        if self.bias is not None or self.activation != 'relu':
            return None
        return self.act_gain * gain, (self.conv_clamp * gain if self.conv_clamp is not None else None)

    def forward(self, x, gain=1, no_act=False):
        b = self.bias.to(x.dtype) if self.bias is not None else None
        if not no_act:
            act_gain = self.act_gain * gain
            act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
            x = bias_act.bias_act(x, b, act=self.activation, gain=act_gain, clamp=act_clamp)
        flip_weight = (self.up == 1)
        return conv2d_resample.conv2d_resample(x=x, w=self.weight.to(x.dtype), f=self.resample_filter, up=self.up, down=self.down,
                                               padding=self.padding, flip_weight=flip_weight, wgain=self.weight_gain)

@persistence.persistent_class
class Spade_Norm_Block(torch.nn.Module):
    """InstanceNorm(x) * (1 + gamma(feat)) + beta(feat) (networks.py:4358-4379)."""
    def __init__(self, in_channels, norm_channels):
        super().__init__()
        self.conv_mlp = Spade_Conv2dLayer(in_channels, norm_channels, kernel_size=3, bias=False)
        self.conv_mlp_act = nn.ReLU()
        self.conv_gamma = Spade_Conv2dLayer(norm_channels, norm_channels, kernel_size=3, bias=False)
        self.conv_beta = Spade_Conv2dLayer(norm_channels, norm_channels, kernel_size=3, bias=False)
        self.param_free_norm = nn.InstanceNorm2d(norm_channels, affine=False)

    def forward(self, x, denorm_feats, post_act=None):
        # conv_mlp (no activation in front, no bias) followed by nn.ReLU (:4373-4374): the ReLU rides in the convolution's epilogue
        m = self.conv_mlp
        actv = conv2d_resample.conv2d_resample_bias_act(x=denorm_feats, w=m.weight.to(denorm_feats.dtype), b=None, f=m.resample_filter,
                                                        up=m.up, down=m.down, padding=m.padding, flip_weight=(m.up == 1),
                                                        act='relu', gain=1, wgain=m.weight_gain)
        # conv_gamma and conv_beta (:4375-4376) read the same tensor: ONE convolution with the concatenated weights writes
        # gamma | beta as channel halves, the normalisation kernel reads (and, backwards, writes) the halves in place
        g, b = self.conv_gamma, self.conv_beta
        same = (g.weight.shape == b.weight.shape and g.up == b.up == 1 and g.down == b.down == 1 and g.padding == b.padding
                and g.bias is None and b.bias is None and g.weight_gain == b.weight_gain and actv.dtype == torch.float32)
        if same:
            gamma = conv2d_resample.conv2d_resample(x=actv, w=torch.cat([g.weight, b.weight], dim=0), f=g.resample_filter,
                                                    padding=g.padding, flip_weight=True, wgain=g.weight_gain)
            beta = None
        else:
            gamma = g(actv, no_act=True)
            beta = b(actv, no_act=True)
        if post_act is not None:        # (relu gain, clamp) of the consuming Spade_Conv2dLayer, applied in the same pass
            return spade_modulate(x, gamma, beta, eps=self.param_free_norm.eps, relu_gain=post_act[0], clamp=post_act[1])
        return spade_modulate(x, gamma, beta, eps=self.param_free_norm.eps)

@persistence.persistent_class
class Spade_ResBlockV2(torch.nn.Module):
    """networks.py:5229-5273"""
    def __init__(self, in_channels, out_channels, kernel_size=3, bias=True, activation='linear', up=1, down=1,
                 resample_filter=[1,3,3,1], conv_clamp=None, channels_last=False, trainable=True, resolution=128):
        super().__init__()
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        common = dict(bias=False, resample_filter=resample_filter, conv_clamp=conv_clamp, channels_last=channels_last)
        self.conv = Spade_Conv2dLayer(in_channels, in_channels, kernel_size=3, **common)
        self.conv0 = Spade_Conv2dLayer(in_channels, out_channels, kernel_size=3, **common)
        self.conv1 = Spade_Conv2dLayer(out_channels, out_channels, kernel_size=3, **common)
        self.skip = Spade_Conv2dLayer(in_channels, out_channels, kernel_size=1, **common)
        feat_channels = 128 * 2 if resolution == 128 else 64 * 2
        self.spade_skip = Spade_Norm_Block(feat_channels, in_channels)
        self.spade0 = Spade_Norm_Block(feat_channels, in_channels)
        self.spade1 = Spade_Norm_Block(feat_channels, out_channels)

    def forward(self, x, denorm_feat):
        x = self.conv(x, no_act=True)
        y = self._normed_conv(self.spade_skip, self.skip, x, denorm_feat, np.sqrt(0.5))
        x = self._normed_conv(self.spade0, self.conv0, x, denorm_feat, 1)
        x = self._normed_conv(self.spade1, self.conv1, x, denorm_feat, np.sqrt(0.5))
        return y.add_(x)

    @staticmethod
    def _normed_conv(norm, conv, x, denorm_feat, gain):
        """conv(norm(x, feat), gain): the activation in front of the convolution is applied by the SPADE kernel when
        the layer allows it."""
        post = conv.fusable_activation(gain)
        if post is None:
            return conv(norm(x, denorm_feat), gain=gain)
        return conv(norm(x, denorm_feat, post_act=post), no_act=True)

#----------------------------------------------------------------------------
# Full-body generator.

@persistence.persistent_class
class SynthesisBlockFull(torch.nn.Module):
    """networks.py:5614-5719"""
    def __init__(self,
        in_channels,                        # Number of input channels, 0 = first block.
        out_channels,                       # Number of output channels.
        w_dim,                              # Intermediate latent (W) dimensionality.
        resolution,                         # Resolution of this block.
        img_channels,                       # Number of output color channels.
        is_last,                            # Is this the last block?
        is_style            = False,        # Is this the block in the sytle synthesis branch
        architecture        = 'skip',       # Architecture: 'orig', 'skip', 'resnet'.
        resample_filter     = [1,3,3,1],    # Low-pass filter to apply when resampling activations.
        conv_clamp          = None,         # Clamp the output of convolution layers to +-X, None = disable clamping.
        use_fp16            = False,        # Use FP16 for this block?
        fp16_channels_last  = False,        # Use channels-last memory format with FP16?
        **layer_kwargs,                     # Arguments for SynthesisLayer.
    ):
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels = in_channels
        self.w_dim = w_dim
        self.resolution = resolution
        self.img_channels = img_channels
        self.is_last = is_last
        self.architecture = architecture
        self.use_fp16 = use_fp16
        self.channels_last = (use_fp16 and fp16_channels_last)
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.num_conv = 0
        self.num_torgb = 0
        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))   # unused: the pose feature replaces it
        if in_channels != 0:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2,
                                        resample_filter=resample_filter, conv_clamp=conv_clamp, channels_last=self.channels_last, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution,
                                    conv_clamp=conv_clamp, channels_last=self.channels_last, **layer_kwargs)
        self.num_conv += 1
        if is_last or architecture == 'skip':
            self.torgb = self._make_torgb(out_channels, img_channels, w_dim, conv_clamp, is_last, is_style)
            self.num_torgb += 1
        if in_channels != 0 and architecture == 'resnet':
            self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=2,
                                    resample_filter=resample_filter, channels_last=self.channels_last)
        if self.resolution > 16:
            self.merge_conv = Conv2dLayer(out_channels + 64, out_channels, kernel_size=1,
                                          resample_filter=resample_filter, channels_last=self.channels_last)

    def _make_torgb(self, out_channels, img_channels, w_dim, conv_clamp, is_last, is_style):
        return ToRGBLayerFull(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp,
                              channels_last=self.channels_last, is_last=is_last, is_style=is_style)

    _num_heads = 1      # extra outputs of the ToRGB layer besides the image (parsing logits)

    def forward(self, x, img, ws, pose_feature, cat_feat, force_fp32=False, fused_modconv=None, **layer_kwargs):
        misc.assert_shape(ws, [None, self.num_conv + self.num_torgb, self.w_dim])
        w_iter = iter(ws.unbind(dim=1))
        dtype = torch.float16 if self.use_fp16 and not force_fp32 else torch.float32
        memory_format = torch.channels_last if self.channels_last and not force_fp32 else torch.contiguous_format
        if fused_modconv is None:
            fused_modconv = (not self.training) and (dtype == torch.float32 or int(x.shape[0]) == 1)

        if self.in_channels == 0:
            x = pose_feature.to(dtype=dtype, memory_format=memory_format)
        else:
            misc.assert_shape(x, [None, self.in_channels, self.resolution // 2, self.resolution // 2])
            x = x.to(dtype=dtype, memory_format=memory_format)

        if self.in_channels == 0:
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
        elif self.architecture == 'resnet':
            y = self.skip(x, gain=np.sqrt(0.5))
            x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, gain=np.sqrt(0.5), **layer_kwargs)
            x = y.add_(x)
        else:
            x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            if x.shape[2] > 16:     # merge the warped-clothing feature of this resolution
                x = torch.cat([x, cat_feat[str(x.shape[2])].to(dtype=dtype, memory_format=memory_format)], dim=1)
                x = self.merge_conv(x)

        heads = (None,) * self._num_heads
        if img is not None:
            misc.assert_shape(img, [None, self.img_channels, self.resolution // 2, self.resolution // 2])
            img = upfirdn2d.upsample2d(img, self.resample_filter)
        if self.is_last or self.architecture == 'skip':
            y, *heads = self.torgb(x, next(w_iter), fused_modconv=fused_modconv)
            y = y.to(dtype=torch.float32, memory_format=torch.contiguous_format)
            img = img.add_(y) if img is not None else y
        return (x, img, *heads)

@persistence.persistent_class
class SynthesisNetworkFull(torch.nn.Module):
    """Style branch b4..b256, parsing-routed SPADE blocks at 128^2 and the texture block (networks.py:5722-5840)."""
    def __init__(self,
        w_dim,                      # Intermediate latent (W) dimensionality.
        img_resolution,             # Output image resolution.
        img_channels,               # Number of color channels.
        channel_base    = 32768,    # Overall multiplier for the number of channels.
        channel_max     = 512,      # Maximum number of channels in any layer.
        num_fp16_res    = 0,        # Use FP16 for the N highest resolutions.
        **block_kwargs,             # Arguments for SynthesisBlock.
    ):
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        super().__init__()
        self.w_dim = w_dim
        self.img_resolution = img_resolution
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.img_channels = img_channels
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        channels_dict = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}

        self.num_ws = 0
        for res in self.block_resolutions:
            in_channels = channels_dict[res // 2] if res > 4 else 0
            out_channels = channels_dict[res]
            is_last = (res == self.img_resolution)
            block = SynthesisBlockFull(in_channels, out_channels, w_dim=w_dim, resolution=res, img_channels=img_channels,
                                       is_last=is_last, is_style=True, use_fp16=False, **block_kwargs)   # fp32 always (networks.py:5747-5748)
            self.num_ws += block.num_conv
            if is_last:
                self.num_ws += block.num_torgb
            setattr(self, f'b{res}', block)

        res = self.block_resolutions[-2]
        self.spade_b128_1 = Spade_ResBlockV2(channels_dict[res], channels_dict[res])
        self.spade_b128_2 = Spade_ResBlockV2(channels_dict[res], channels_dict[res])
        self.spade_b128_3 = Spade_ResBlockV2(channels_dict[res], channels_dict[res])

        res = self.block_resolutions[-1]
        self.texture_b256 = SynthesisBlockFull(channels_dict[res // 2], channels_dict[res], w_dim=w_dim, resolution=res,
                                               img_channels=img_channels, is_last=True, is_style=False, use_fp16=False, **block_kwargs)
        ngf = 64
        self.spade_encoder = nn.Sequential(
            Conv2dLayer(3, ngf, kernel_size=7, activation='relu'),
            ResBlock(ngf, ngf, kernel_size=4, activation='relu'),
            ResBlock(ngf, ngf * 2, kernel_size=4, activation='relu', down=2))

    def get_spade_feat(self, mask_256, denorm_mask, denorm_input):
        """Clothing features at 128^2; where the predicted region is not covered by the warped
        clothing, fill with the masked mean feature (networks.py:5777-5800)."""
        dt = mask_256.dtype
        mask_256 = (mask_256 > 0.9).to(dt)
        mask_128 = (torch.nn.functional.interpolate(mask_256, scale_factor=0.5) > 0.9).to(dt)
        denorm_mask_128 = (torch.nn.functional.interpolate(denorm_mask, scale_factor=0.5) > 0.9).to(dt)
        valid_mask = ((mask_128 + denorm_mask_128) == 2.0).to(dt)
        res_mask = mask_128 - valid_mask

        denorm_input = denorm_input * mask_256 - (1 - mask_256)
        feat = self.spade_encoder(denorm_input)
        valid_feat_sum = torch.sum(feat * valid_mask, dim=(2, 3), keepdim=True)
        valid_mask_sum = torch.sum(valid_mask, dim=(2, 3), keepdim=True)
        valid_index = (valid_mask_sum > 10).to(dt)
        valid_mask_sum = valid_mask_sum * valid_index + (128 * 128) * (1 - valid_index)
        average_feat = valid_feat_sum / valid_mask_sum
        return feat * (1 - res_mask) + average_feat * res_mask

    def forward(self, ws, pose_feat, cat_feat, denorm_upper_input, denorm_lower_input, denorm_upper_mask,
                denorm_lower_mask, **block_kwargs):
        misc.assert_shape(ws, [None, self.num_ws, self.w_dim])
        ws = ws.to(torch.float32)
        block_ws = []
        w_idx = 0
        for res in self.block_resolutions:
            block = getattr(self, f'b{res}')
            block_ws.append(ws.narrow(1, w_idx, block.num_conv + block.num_torgb))
            w_idx += block.num_conv

        x = img = pred_parsing = None
        for res, cur_ws in zip(self.block_resolutions, block_ws):
            block = getattr(self, f'b{res}')
            x, img, pred_parsing = block(x, img, cur_ws, pose_feat, cat_feat, force_fp32=True, **block_kwargs)
            if res == 128:
                x_128, img_128 = x.clone(), img.clone()

        parsing_index = torch.argmax(torch.softmax(pred_parsing.detach(), dim=1), dim=1)[:, None, ...].float()
        upper_mask = (parsing_index == 1).float()
        lower_mask = (parsing_index == 2).float()
        spade_upper_feat = self.get_spade_feat(upper_mask.detach(), denorm_upper_mask, denorm_upper_input)
        spade_lower_feat = self.get_spade_feat(lower_mask.detach(), denorm_lower_mask, denorm_lower_input)
        spade_feat = torch.cat([spade_upper_feat, spade_lower_feat], dim=1)

        x_spade_128 = self.spade_b128_1(x_128, spade_feat)
        x_spade_128 = self.spade_b128_2(x_spade_128, spade_feat)
        x_spade_128 = self.spade_b128_3(x_spade_128, spade_feat)

        _, finetune_img, _ = self.texture_b256(x_spade_128, img_128, block_ws[-1], pose_feat, cat_feat, force_fp32=True, **block_kwargs)
        return img, finetune_img, pred_parsing

@persistence.persistent_class
class GeneratorFull(torch.nn.Module):
    """Pose encoder + patch style encoder + mapping + synthesis (networks.py:5843-5881)."""
    def __init__(self,
        z_dim,                      # Input latent (Z) dimensionality.
        c_dim,                      # Conditioning label (C) dimensionality.
        w_dim,                      # Intermediate latent (W) dimensionality.
        img_resolution,             # Output resolution.
        img_channels,               # Number of output color channels.
        mapping_kwargs      = {},   # Arguments for MappingNetwork.
        synthesis_kwargs    = {},   # Arguments for SynthesisNetwork.
    ):
        super().__init__()
        self.z_dim = z_dim
        self.c_dim = c_dim
        self.w_dim = w_dim
        self.img_resolution = img_resolution
        self.img_channels = img_channels
        self.synthesis = SynthesisNetworkFull(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)
        self.const_encoding = ConstEncoderNetwork(input_nc=3 + 3, output_nc=512, ngf=64, n_downsampling=6)
        self.style_encoding = StyleEncoderNetworkV16(input_nc=(10 * 3 + 4 * 3), output_nc=512, ngf=64, n_downsampling=6)

    def forward(self, z, c, retain, pose, denorm_upper_input, denorm_lower_input, denorm_upper_mask, denorm_lower_mask,
                truncation_psi=1, truncation_cutoff=None, **synthesis_kwargs):
        pose_feat = self.const_encoding(pose)
        stylecode, feats = self.style_encoding(c, retain)
        ws = self.mapping(z, stylecode, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        cat_feats = {str(feat.shape[2]): feat for feat in feats}
        return self.synthesis(ws, pose_feat, cat_feats, denorm_upper_input, denorm_lower_input,
                              denorm_upper_mask, denorm_lower_mask, **synthesis_kwargs)

#----------------------------------------------------------------------------
# The released 256x192 inference model (test.py): same skeleton, sigmoid mask heads instead of parsing logits.

@persistence.persistent_class
class SynthesisBlockV18(SynthesisBlockFull.__mro__[1]):
    """networks.py:5313-5418: SynthesisBlockFull with ToRGBLayerV18 (returns x, img, upper_mask, lower_mask)."""
    _num_heads = 2

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, **kwargs):
        super().__init__(in_channels, out_channels, w_dim, resolution, img_channels, is_last, **kwargs)

    def _make_torgb(self, out_channels, img_channels, w_dim, conv_clamp, is_last, is_style):
        return ToRGBLayerV18(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp, channels_last=self.channels_last, is_last=is_last)

@persistence.persistent_class
class SynthesisNetworkV18(torch.nn.Module):
    """networks.py:5419-5531"""
    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=0, **block_kwargs):
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        super().__init__()
        self.w_dim = w_dim
        self.img_resolution = img_resolution
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.img_channels = img_channels
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        channels_dict = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        self.num_ws = 0
        for res in self.block_resolutions:
            in_channels = channels_dict[res // 2] if res > 4 else 0
            is_last = (res == self.img_resolution)
            block = SynthesisBlockV18(in_channels, channels_dict[res], w_dim=w_dim, resolution=res, img_channels=img_channels,
                                      is_last=is_last, use_fp16=False, **block_kwargs)
            self.num_ws += block.num_conv
            if is_last:
                self.num_ws += block.num_torgb
            setattr(self, f'b{res}', block)
        res = self.block_resolutions[-2]
        self.spade_b128_1 = Spade_ResBlockV2(channels_dict[res], channels_dict[res])
        self.spade_b128_2 = Spade_ResBlockV2(channels_dict[res], channels_dict[res])
        self.spade_b128_3 = Spade_ResBlockV2(channels_dict[res], channels_dict[res])
        res = self.block_resolutions[-1]
        self.texture_b256 = SynthesisBlockV18(channels_dict[res // 2], channels_dict[res], w_dim=w_dim, resolution=res,
                                              img_channels=img_channels, is_last=True, use_fp16=False, **block_kwargs)
        ngf = 64
        self.spade_encoder = nn.Sequential(
            Conv2dLayer(3, ngf, kernel_size=7, activation='relu'),
            ResBlock(ngf, ngf, kernel_size=4, activation='relu'),
            ResBlock(ngf, ngf * 2, kernel_size=4, activation='relu', down=2))

    get_spade_feat = SynthesisNetworkFull.__mro__[1].get_spade_feat

    def forward(self, ws, pose_feat, cat_feat, denorm_upper_input, denorm_lower_input, denorm_upper_mask, denorm_lower_mask, **block_kwargs):
        misc.assert_shape(ws, [None, self.num_ws, self.w_dim])
        ws = ws.to(torch.float32)
        block_ws = []
        w_idx = 0
        for res in self.block_resolutions:
            block = getattr(self, f'b{res}')
            block_ws.append(ws.narrow(1, w_idx, block.num_conv + block.num_torgb))
            w_idx += block.num_conv
        x = img = upper_mask = lower_mask = None
        for res, cur_ws in zip(self.block_resolutions, block_ws):
            x, img, upper_mask, lower_mask = getattr(self, f'b{res}')(x, img, cur_ws, pose_feat, cat_feat, force_fp32=True, **block_kwargs)
            if res == 128:
                x_128, img_128 = x.clone(), img.clone()
        spade_feat = torch.cat([self.get_spade_feat(upper_mask.detach(), denorm_upper_mask, denorm_upper_input),
                                self.get_spade_feat(lower_mask.detach(), denorm_lower_mask, denorm_lower_input)], dim=1)
        x_spade_128 = self.spade_b128_1(x_128, spade_feat)
        x_spade_128 = self.spade_b128_2(x_spade_128, spade_feat)
        x_spade_128 = self.spade_b128_3(x_spade_128, spade_feat)
        _, finetune_img, _, _ = self.texture_b256(x_spade_128, img_128, block_ws[-1], pose_feat, cat_feat, force_fp32=True, **block_kwargs)
        return img, finetune_img, upper_mask, lower_mask

@persistence.persistent_class
class GeneratorV18(torch.nn.Module):
    """networks.py:5534-5577 (the class test.py's pretrained pickle instantiates; 60-channel patch input)."""
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, synthesis_kwargs={}):
        super().__init__()
        self.z_dim = z_dim
        self.c_dim = c_dim
        self.w_dim = w_dim
        self.img_resolution = img_resolution
        self.img_channels = img_channels
        self.synthesis = SynthesisNetworkV18(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)
        self.const_encoding = ConstEncoderNetwork(input_nc=3 + 3, output_nc=512, ngf=64, n_downsampling=6)
        self.style_encoding = StyleEncoderNetworkV16(input_nc=30 * 2, output_nc=512, ngf=64, n_downsampling=6)

    def forward(self, z, c, retain, pose, denorm_upper_input, denorm_lower_input, denorm_upper_mask, denorm_lower_mask,
                truncation_psi=1, truncation_cutoff=None, **synthesis_kwargs):
        pose_feat = self.const_encoding(pose)
        stylecode, feats = self.style_encoding(c, retain)
        ws = self.mapping(z, stylecode, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        cat_feats = {str(feat.shape[2]): feat for feat in feats}
        return self.synthesis(ws, pose_feat, cat_feats, denorm_upper_input, denorm_lower_input,
                              denorm_upper_mask, denorm_lower_mask, **synthesis_kwargs)

#----------------------------------------------------------------------------
# Discriminator.

@persistence.persistent_class
class DiscriminatorBlock(torch.nn.Module):
    """fromrgb (first block) -> [3x3, 3x3 /2] + 1x1 /2 skip (networks.py:916-996)."""
    def __init__(self,
        in_channels,                        # Number of input channels, 0 = first block.
        tmp_channels,                       # Number of intermediate channels.
        out_channels,                       # Number of output channels.
        resolution,                         # Resolution of this block.
        img_channels,                       # Number of input color channels.
        first_layer_idx,                    # Index of the first layer.
        architecture        = 'resnet',     # Architecture: 'orig', 'skip', 'resnet'.
        activation          = 'lrelu',      # Activation function: 'relu', 'lrelu', etc.
        resample_filter     = [1,3,3,1],    # Low-pass filter to apply when resampling activations.
        conv_clamp          = None,         # Clamp the output of convolution layers to +-X, None = disable clamping.
        use_fp16            = False,        # Use FP16 for this block?
        fp16_channels_last  = False,        # Use channels-last memory format with FP16?
        freeze_layers       = 0,            # Freeze-D: Number of layers to freeze.
    ):
        assert in_channels in [0, tmp_channels]
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels = in_channels
        self.resolution = resolution
        self.img_channels = img_channels
        self.first_layer_idx = first_layer_idx
        self.architecture = architecture
        self.use_fp16 = use_fp16
        self.channels_last = (use_fp16 and fp16_channels_last)
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))

        self.num_layers = 0
        def next_trainable():
            trainable = (self.first_layer_idx + self.num_layers >= freeze_layers)
            self.num_layers += 1
            return trainable

        if in_channels == 0 or architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, tmp_channels, kernel_size=1, activation=activation,
                                       trainable=next_trainable(), conv_clamp=conv_clamp, channels_last=self.channels_last)
        self.conv0 = Conv2dLayer(tmp_channels, tmp_channels, kernel_size=3, activation=activation,
                                 trainable=next_trainable(), conv_clamp=conv_clamp, channels_last=self.channels_last)
        self.conv1 = Conv2dLayer(tmp_channels, out_channels, kernel_size=3, activation=activation, down=2,
                                 trainable=next_trainable(), resample_filter=resample_filter, conv_clamp=conv_clamp, channels_last=self.channels_last)
        if architecture == 'resnet':
            self.skip = Conv2dLayer(tmp_channels, out_channels, kernel_size=1, bias=False, down=2,
                                    trainable=next_trainable(), resample_filter=resample_filter, channels_last=self.channels_last)

    def forward(self, x, img, force_fp32=False):
        dtype = torch.float16 if self.use_fp16 and not force_fp32 else torch.float32
        memory_format = torch.channels_last if self.channels_last and not force_fp32 else torch.contiguous_format
        if x is not None:
            misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
            x = x.to(dtype=dtype, memory_format=memory_format)
        if self.in_channels == 0 or self.architecture == 'skip':
            misc.assert_shape(img, [None, self.img_channels, self.resolution, self.resolution])
            img = img.to(dtype=dtype, memory_format=memory_format)
            y = self.fromrgb(img)
            x = x + y if x is not None else y
            img = upfirdn2d.downsample2d(img, self.resample_filter) if self.architecture == 'skip' else None
        if self.architecture == 'resnet':
            y = self.skip(x, gain=np.sqrt(0.5))
            x = self.conv0(x)
            x = self.conv1(x, gain=np.sqrt(0.5))
            x = y.add_(x)
        else:
            x = self.conv0(x)
            x = self.conv1(x)
        assert x.dtype == dtype
        return x, img

@persistence.persistent_class
class MinibatchStdLayer(torch.nn.Module):
    """Append the per-group feature standard deviation as extra channels (networks.py:1000-1022)."""
    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size = group_size
        self.num_channels = num_channels

    def forward(self, x):
        N, C, H, W = x.shape
        G = min(int(self.group_size), int(N)) if self.group_size is not None else int(N)
        F = self.num_channels
        c = C // F
        y = x.reshape(G, -1, F, c, H, W)
        y = y - y.mean(dim=0)
        y = y.square().mean(dim=0)
        y = (y + 1e-8).sqrt()
        y = y.mean(dim=[2, 3, 4])
        y = y.reshape(-1, F, 1, 1).repeat(G, 1, H, W)
        return torch.cat([x, y], dim=1)

@persistence.persistent_class
class DiscriminatorEpilogue(torch.nn.Module):
    """mbstd -> 3x3 -> FC -> FC -> projection on the conditioning vector (networks.py:1026-1080)."""
    def __init__(self,
        in_channels,                    # Number of input channels.
        cmap_dim,                       # Dimensionality of mapped conditioning label, 0 = no label.
        resolution,                     # Resolution of this block.
        img_channels,                   # Number of input color channels.
        architecture        = 'resnet', # Architecture: 'orig', 'skip', 'resnet'.
        mbstd_group_size    = 4,        # Group size for the minibatch standard deviation layer, None = entire minibatch.
        mbstd_num_channels  = 1,        # Number of features for the minibatch standard deviation layer, 0 = disable.
        activation          = 'lrelu',  # Activation function: 'relu', 'lrelu', etc.
        conv_clamp          = None,     # Clamp the output of convolution layers to +-X, None = disable clamping.
    ):
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels = in_channels
        self.cmap_dim = cmap_dim
        self.resolution = resolution
        self.img_channels = img_channels
        self.architecture = architecture
        if architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, in_channels, kernel_size=1, activation=activation)
        self.mbstd = MinibatchStdLayer(group_size=mbstd_group_size, num_channels=mbstd_num_channels) if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation, conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * (resolution ** 2), in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, 1 if cmap_dim == 0 else cmap_dim)

    def forward(self, x, img, cmap, force_fp32=False):
        misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
        _ = force_fp32
        x = x.to(dtype=torch.float32, memory_format=torch.contiguous_format)
        if self.architecture == 'skip':
            misc.assert_shape(img, [None, self.img_channels, self.resolution, self.resolution])
            x = x + self.fromrgb(img.to(dtype=torch.float32, memory_format=torch.contiguous_format))
        if self.mbstd is not None:
            x = self.mbstd(x)
        x = self.conv(x)
        x = self.fc(x.flatten(1))
        x = self.out(x)
        if self.cmap_dim > 0:
            misc.assert_shape(cmap, [None, self.cmap_dim])
            x = (x * cmap).sum(dim=1, keepdim=True) * (1 / np.sqrt(self.cmap_dim))
        assert x.dtype == torch.float32
        return x

@persistence.persistent_class
class Discriminator(torch.nn.Module):
    """Residual StyleGAN2 discriminator with projection conditioning (networks.py:1084-1139)."""
    def __init__(self,
        c_dim,                          # Conditioning label (C) dimensionality.
        img_resolution,                 # Input resolution.
        img_channels,                   # Number of input color channels.
        architecture        = 'resnet', # Architecture: 'orig', 'skip', 'resnet'.
        channel_base        = 32768,    # Overall multiplier for the number of channels.
        channel_max         = 512,      # Maximum number of channels in any layer.
        num_fp16_res        = 0,        # Use FP16 for the N highest resolutions.
        conv_clamp          = None,     # Clamp the output of convolution layers to +-X, None = disable clamping.
        cmap_dim            = None,     # Dimensionality of mapped conditioning label, None = default.
        block_kwargs        = {},       # Arguments for DiscriminatorBlock.
        mapping_kwargs      = {},       # Arguments for MappingNetwork.
        epilogue_kwargs     = {},       # Arguments for DiscriminatorEpilogue.
    ):
        super().__init__()
        self.c_dim = c_dim
        self.img_resolution = img_resolution
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.img_channels = img_channels
        self.block_resolutions = [2 ** i for i in range(self.img_resolution_log2, 2, -1)]
        channels_dict = {res: min(channel_base // res, channel_max) for res in self.block_resolutions + [4]}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)
        if cmap_dim is None:
            cmap_dim = channels_dict[4]
        if c_dim == 0:
            cmap_dim = 0
        common_kwargs = dict(img_channels=img_channels, architecture=architecture, conv_clamp=conv_clamp)
        cur_layer_idx = 0
        for res in self.block_resolutions:
            in_channels = channels_dict[res] if res < img_resolution else 0
            block = DiscriminatorBlock(in_channels, channels_dict[res], channels_dict[res // 2], resolution=res,
                                       first_layer_idx=cur_layer_idx, use_fp16=(res >= fp16_resolution), **block_kwargs, **common_kwargs)
            setattr(self, f'b{res}', block)
            cur_layer_idx += block.num_layers
        if c_dim > 0:
            self.mapping = MappingNetwork(z_dim=0, c_dim=c_dim, w_dim=cmap_dim, num_ws=None, w_avg_beta=None, **mapping_kwargs)
        self.b4 = DiscriminatorEpilogue(channels_dict[4], cmap_dim=cmap_dim, resolution=4, **epilogue_kwargs, **common_kwargs)

    def forward(self, img, c, **block_kwargs):
        x = None
        for res in self.block_resolutions:
            x, img = getattr(self, f'b{res}')(x, img, **block_kwargs)
        cmap = self.mapping(None, c) if self.c_dim > 0 else None
        return self.b4(x, img, cmap)

#----------------------------------------------------------------------------
