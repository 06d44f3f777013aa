"""Fused bias + activation + gain + clamp on the MI355X HIP kernel.

Public surface mirrors the reference module torch_utils/ops/bias_act.py
(``activation_funcs`` :23-33, ``bias_act`` :55-89). First and second derivatives are the
same kernel in its ``grad=1`` / ``grad=2`` modes, arranged as in the reference's
``BiasActCuda`` / ``BiasActCudaGrad`` pair (:145-206). The bias gradient is reduced by
``pasta_bias_grad`` instead of a separate ``sum`` pass through PyTorch.
"""

import numpy as np
import torch
import dnnlib

from . import _native

#----------------------------------------------------------------------------

def _spec(func, def_alpha, def_gain, cuda_idx, ref, has_2nd_grad):
    return dnnlib.EasyDict(func=func, def_alpha=def_alpha, def_gain=def_gain, cuda_idx=cuda_idx, ref=ref, has_2nd_grad=has_2nd_grad)

# name -> properties; ``cuda_idx`` is the activation code of the native kernel, ``ref`` says which
# forward tensor the derivative is expressed in ('x', 'y' or '' for none).
activation_funcs = {
    'linear':   _spec(lambda x, **_: x,                                          0,   1,          1, '',  False),
    'relu':     _spec(lambda x, **_: torch.nn.functional.relu(x),                0,   np.sqrt(2), 2, 'y', False),
    'lrelu':    _spec(lambda x, alpha, **_: torch.nn.functional.leaky_relu(x, alpha), 0.2, np.sqrt(2), 3, 'y', False),
    'tanh':     _spec(lambda x, **_: torch.tanh(x),                              0,   1,          4, 'y', True),
    'sigmoid':  _spec(lambda x, **_: torch.sigmoid(x),                           0,   1,          5, 'y', True),
    'elu':      _spec(lambda x, **_: torch.nn.functional.elu(x),                 0,   1,          6, 'y', True),
    'selu':     _spec(lambda x, **_: torch.nn.functional.selu(x),                0,   1,          7, 'y', True),
    'softplus': _spec(lambda x, **_: torch.nn.functional.softplus(x),            0,   1,          8, 'y', True),
    'swish':    _spec(lambda x, **_: torch.sigmoid(x) * x,                       0,   np.sqrt(2), 9, 'x', True),
}

#----------------------------------------------------------------------------
# Slope tape: a TEST instrument (tests/test_training_step_gpu.py; None = off, one ``is not None`` per activation otherwise).
# Two evaluations of the same network in different launch plans differ by fp32 rounding, and a leaky-ReLU / ReLU pre-activation within
# rounding of zero then takes the other slope in one of them: one unit's share of ONE sample's gradient moves by percent while everything else
# agrees to 1e-6 -- which makes "equal gradients" untestable at fp32 accuracy, and which unit flips changes with every rounding anywhere
# upstream.  The tape removes the flips from such a comparison: in ``record`` mode it notes the sign mask of every piecewise-linear
# activation output in call order (the fused convolution epilogues included: conv2d_gradfix._ConvBiasActHip); in ``replay`` mode the
# outputs whose sign differs from the recorded one are moved across zero (to 1e-30 where the recorded pass took the positive slope, to 0
# where it took the other: the backward kernels read the slope off the sign of the saved output), so both passes differentiate the SAME
# piecewise-linear function.  ``moved`` counts the elements and ``worst`` keeps the largest magnitude moved, relative to the tensor's
# maximum -- the test asserts it is rounding-sized.

class SlopeTape:
    def __init__(self, replay=None, select=None):
        self.masks = [] if replay is None else None
        self.replay, self.select, self.pos, self.moved, self.worst = replay, select, 0, 0, 0.0

    def visit(self, y, act):
        if act not in ('lrelu', 'relu') or y.numel() == 0:
            return y
        if self.masks is not None:
            self.masks.append(y > 0)
            return y
        want = self.replay[self.pos]
        self.pos += 1
        if self.select is not None:
            want = self.select(want)
        assert want.shape == y.shape, (self.pos - 1, tuple(want.shape), tuple(y.shape))
        flip = (y > 0) != want
        n = int(flip.sum())
        if n:
            self.moved += n
            self.worst = max(self.worst, float(y[flip].abs().max() / y.abs().max()))
            y.masked_fill_(flip & want, 1e-30).masked_fill_(flip & ~want, 0.0)
        return y

slope_tape = None

#----------------------------------------------------------------------------

def _dense_format(t):
    """Memory format the kernel will run in (the tensor must be dense in that order)."""
    if t.ndim > 2 and t.stride(1) == 1 and t.is_contiguous(memory_format=torch.channels_last):
        return torch.channels_last
    return torch.contiguous_format

def _launch(x, b, xref, yref, dy, grad, dim, act_idx, alpha, gain, clamp):
    """One ``pasta_bias_act`` launch; every tensor shares x's dense layout, ``None`` = absent."""
    _native.require_gpu(x, 'bias_act')
    for name, t in (('xref', xref), ('yref', yref), ('dy', dy)):
        if t is not None and (t.shape != x.shape or t.dtype != x.dtype or t.device != x.device or t.stride() != x.stride()):
            raise RuntimeError(f'bias_act: {name} must have the same shape, dtype, device and layout as x')
    size_b, step_b = 1, 1
    if b is not None:
        if b.ndim != 1:
            raise RuntimeError('bias_act: b must have rank 1')
        if b.dtype != x.dtype or b.device != x.device:
            raise RuntimeError('bias_act: b must have the same dtype and device as x')
        if not (0 <= dim < x.ndim):
            raise RuntimeError('bias_act: dim is out of bounds')
        if b.numel() != x.shape[dim]:
            raise RuntimeError('bias_act: b has wrong number of elements')
        size_b, step_b = b.numel(), x.stride(dim)
    y = torch.empty_like(x)
    if x.numel() == 0:
        return y
    row = _native.amax_slot(y) if x.numel() >= 1 << 16 else None      # feature maps: the convolution after this one wants |max|
    with torch.cuda.device(x.device):
        st = _native.lib().pasta_bias_act(
            _native.ptr(x), _native.ptr(b), _native.ptr(xref), _native.ptr(yref), _native.ptr(dy), _native.ptr(y),
            _native.dtype_code(x, 'bias_act'), x.numel(), size_b, step_b, grad, act_idx,
            float(alpha), float(gain), float(clamp), _native.stream(), _native.ptr(row))
    _native.check(st)
    return _native.amax_attach(y, row)

def _bias_grad(dx, dim):
    """Sum ``dx`` over every dimension but ``dim`` with the native two-stage reduction."""
    size_b, step_b = dx.shape[dim], dx.stride(dim)
    lib = _native.lib()
    db = torch.empty([size_b], dtype=dx.dtype, device=dx.device)
    if dx.numel() == 0:
        return db.zero_()
    nbytes = lib.pasta_bias_grad_workspace(dx.numel(), size_b, step_b)
    work = torch.empty([max(nbytes // 4, 1)], dtype=torch.float32, device=dx.device)
    with torch.cuda.device(dx.device):
        st = lib.pasta_bias_grad(_native.ptr(dx), _native.ptr(db), _native.ptr(work), _native.dtype_code(dx, 'bias_grad'),
                                 dx.numel(), size_b, step_b, _native.stream())
    _native.check(st)
    return db

def _grad_db_workspace(dy, dim, act_idx):
    """Bytes of scratch for the fused (dx, db) launch, 0 when the case is not covered by it."""
    if dy.device.type != 'cuda' or dy.dtype not in (torch.float32, torch.float16, torch.bfloat16) or not dy.is_contiguous() or dy.numel() == 0:
        return 0
    return _native.lib().pasta_bias_act_grad_db_workspace(_native.dtype_code(dy, 'bias_act'), dy.numel(), dy.shape[dim], dy.stride(dim), act_idx)

def _launch_grad_db(dy, y, dim, act_idx, alpha, gain, clamp, nbytes):
    """dx and db = sum(dx) in one ``pasta_bias_act_grad_db`` launch (dy, y contiguous NCHW)."""
    if y is not None and (y.shape != dy.shape or y.dtype != dy.dtype or y.stride() != dy.stride()):
        raise RuntimeError('bias_act: yref must have the same shape, dtype and layout as dy')
    dx = torch.empty_like(dy)
    db = torch.empty([dy.shape[dim]], dtype=dy.dtype, device=dy.device)
    work = torch.empty([nbytes // 4], dtype=torch.float32, device=dy.device)
    row = _native.amax_slot(dx)
    with torch.cuda.device(dy.device):
        st = _native.lib().pasta_bias_act_grad_db(
            _native.ptr(dy), _native.ptr(y), _native.ptr(dx), _native.ptr(db), _native.ptr(work), _native.dtype_code(dy, 'bias_act'),
            dy.numel(), dy.shape[dim], dy.stride(dim), act_idx, float(alpha), float(gain), float(clamp), _native.stream(), _native.ptr(row))
    _native.check(st)
    return _native.amax_attach(dx, row), db

def grad_with_bias_grad(dy, y, cfg):
    """(dx, db) of ``bias_act`` for the piecewise-linear activations whose derivative is expressed in y
    (``cfg`` = (dim, act, alpha, gain, clamp)); one fused launch when the native kernel covers the case."""
    dim, act, alpha, gain, clamp = cfg
    spec = activation_funcs[act]
    if not spec.has_2nd_grad and 'x' not in spec.ref:
        nbytes = _grad_db_workspace(dy, dim, spec.cuda_idx)
        if nbytes > 0:
            return _BiasActHipGradDb.apply(dy, y, cfg)
    dx = _BiasActHipGrad.apply(dy, None, None, y, cfg)
    return dx, _BiasSum.apply(dx, dim)

def _bias_grad_supported(t, dim):
    """The native reduction views t as [outer, size_b, step_b] with long contiguous step_b runs
    (NCHW feature maps); short runs (FC outputs, channels_last) go through ``Tensor.sum``."""
    return t.numel() > 0 and t.is_contiguous() and t.stride(dim) >= 32

#----------------------------------------------------------------------------

class _BiasActHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, cfg):
        dim, act, alpha, gain, clamp = cfg
        spec = activation_funcs[act]
        fmt = _dense_format(x)
        x = x.contiguous(memory_format=fmt)
        b = b.contiguous() if b is not None else None
        y = x
        if act != 'linear' or gain != 1 or clamp >= 0 or b is not None:
            y = _launch(x, b, None, None, None, 0, dim, spec.cuda_idx, alpha, gain, clamp)
            if slope_tape is not None:
                y = slope_tape.visit(y, act)
        keep_x = 'x' in spec.ref or spec.has_2nd_grad
        # The clamp mask of the gradient is taken from y. The reference's CUDA wrapper drops y for
        # 'linear' (bias_act.py:160) and so lets gradients through a clamped ToRGB output; its CPU
        # path (bias_act.py:121-122, torch.clamp) does not. The CPU path is the parity target.
        keep_y = 'y' in spec.ref or (clamp >= 0 and 'x' not in spec.ref)
        ctx.save_for_backward(x if keep_x else None, b if keep_x else None, y if keep_y else None)
        ctx.cfg = cfg
        ctx.fmt = fmt
        return y

    @staticmethod
    def backward(ctx, dy):
        dim, act, alpha, gain, clamp = ctx.cfg
        x, b, y = ctx.saved_tensors
        dy = dy.contiguous(memory_format=ctx.fmt)
        dx = db = None
        has_kernel = act != 'linear' or gain != 1 or clamp >= 0
        if ctx.needs_input_grad[1] and has_kernel and x is None and ctx.fmt == torch.contiguous_format:
            dx, db = grad_with_bias_grad(dy, y, ctx.cfg)
            return dx, db, None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = dy
            if has_kernel:
                dx = _BiasActHipGrad.apply(dy, x, b, y, ctx.cfg)
        if ctx.needs_input_grad[1]:
            db = _BiasSum.apply(dx, dim)
        return dx, db, None

class _BiasActHipGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, b, y, cfg):
        dim, act, alpha, gain, clamp = cfg
        spec = activation_funcs[act]
        ctx.fmt = _dense_format(dy)
        dx = _launch(dy, b, x, y, None, 1, dim, spec.cuda_idx, alpha, gain, clamp)
        ctx.save_for_backward(dy if spec.has_2nd_grad else None, x, b, y)
        ctx.cfg = cfg
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dim, act, alpha, gain, clamp = ctx.cfg
        spec = activation_funcs[act]
        d_dx = d_dx.contiguous(memory_format=ctx.fmt)
        dy, x, b, y = ctx.saved_tensors
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActHipGrad.apply(d_dx, x, b, y, ctx.cfg)
        if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _launch(d_dx, b, x, y, dy, 2, dim, spec.cuda_idx, alpha, gain, clamp)
        if spec.has_2nd_grad and ctx.needs_input_grad[2]:
            d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
        return d_dy, d_x, d_b, None, None

class _BiasActHipGradDb(torch.autograd.Function):
    """(dx, db) in one pass for linear / relu / lrelu. Both outputs are linear in dy, so the gradient of a
    functional of (dx, db) with respect to dy is the same derivative kernel applied to d_dx + broadcast(d_db)."""
    @staticmethod
    def forward(ctx, dy, y, cfg):
        dim, act, alpha, gain, clamp = cfg
        spec = activation_funcs[act]
        nbytes = _grad_db_workspace(dy, dim, spec.cuda_idx)
        dx, db = _launch_grad_db(dy, y, dim, spec.cuda_idx, alpha, gain, clamp, nbytes)
        ctx.save_for_backward(y)
        ctx.cfg = cfg
        ctx.shape = dy.shape
        ctx.set_materialize_grads(False)
        return dx, db

    @staticmethod
    def backward(ctx, d_dx, d_db):
        dim = ctx.cfg[0]
        y, = ctx.saved_tensors
        if not ctx.needs_input_grad[0] or (d_dx is None and d_db is None):
            return None, None, None
        g = d_dx
        if d_db is not None:
            bc = d_db.reshape([-1 if i == dim else 1 for i in range(len(ctx.shape))]).expand(ctx.shape)
            g = bc if g is None else g + bc
        return _BiasActHipGrad.apply(g.contiguous(), None, None, y, ctx.cfg), None, None

class _BiasSum(torch.autograd.Function):
    """db = sum of dx over all dimensions except ``dim`` (differentiable: the transpose is a broadcast)."""
    @staticmethod
    def forward(ctx, dx, dim):
        ctx.dim = dim
        ctx.shape = dx.shape
        if dx.device.type == 'cuda' and _bias_grad_supported(dx, dim):
            return _bias_grad(dx, dim)
        return dx.sum([i for i in range(dx.ndim) if i != dim])

    @staticmethod
    def backward(ctx, g):
        view = [-1 if i == ctx.dim else 1 for i in range(len(ctx.shape))]
        return g.reshape(view).expand(ctx.shape), None

#----------------------------------------------------------------------------

def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """``clamp(act(x + b) * gain)`` in one pass (reference: bias_act.py:55-89).

    ``b`` is a 1-D tensor matching ``x.shape[dim]`` or ``None``; ``alpha``/``gain`` default to the
    activation's entry in :data:`activation_funcs`; ``clamp=None`` disables clamping. Supports first
    and second order gradients. ``impl='ref'`` is not provided by this package."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'ref':
        raise NotImplementedError("bias_act(impl='ref'): this package has no PyTorch-op fallback; "
                                  "the CPU restatement used for testing is oracle/ref_ops.py")
    assert clamp is None or clamp >= 0
    spec = activation_funcs[act]
    alpha = float(alpha if alpha is not None else spec.def_alpha)
    gain = float(gain if gain is not None else spec.def_gain)
    clamp = float(clamp if clamp is not None else -1)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1
        assert 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
    return _BiasActHip.apply(x, b, (dim, act, alpha, gain, clamp))

#----------------------------------------------------------------------------
