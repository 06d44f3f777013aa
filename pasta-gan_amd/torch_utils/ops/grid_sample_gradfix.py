"""``grid_sample`` with a second derivative (reference: torch_utils/ops/grid_sample_gradfix.py:22-83).

ADA's geometric step resamples the discriminator's input under a per-sample affine grid, and R1 differentiates
``D(augment(real))`` twice with respect to the image, so the image gradient of the bilinear sampler must itself be
differentiable.  PyTorch's ``grid_sampler_2d_backward`` is not.  Bilinear sampling with zero padding is a linear map of
the image for a fixed grid: ``y = S(grid) x``.  Its gradient is ``S^T dy`` (ATen's backward kernel, image part only) and
the gradient of THAT with respect to ``dy`` is ``S`` again, i.e. the forward sampler -- two Functions that call each
other.  The grid never needs a gradient here (it comes from sampled parameters); asking for one is an error, as in the
reference (:75-77).
"""

import torch

enabled = False  # kept for API compatibility: the differentiable path is always on

def grid_sample(input, grid):
    return _Sample.apply(input, grid)

def _sampler(x, grid):
    return torch.nn.functional.grid_sample(input=x, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)

class _Sample(torch.autograd.Function):
    """y = S(grid) x"""
    @staticmethod
    def forward(ctx, x, grid):
        assert x.ndim == 4 and grid.ndim == 4
        ctx.save_for_backward(grid)
        ctx.x_shape = x.shape
        return _sampler(x, grid)

    @staticmethod
    def backward(ctx, dy):
        grid, = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise NotImplementedError('grid_sample_gradfix: no gradient with respect to the sampling grid')
        dx = _SampleAdjoint.apply(dy, grid, ctx.x_shape) if ctx.needs_input_grad[0] else None
        return dx, None

class _SampleAdjoint(torch.autograd.Function):
    """dx = S(grid)^T dy"""
    @staticmethod
    def forward(ctx, dy, grid, x_shape):
        ctx.save_for_backward(grid)
        # the backward kernel only reads the image's shape / dtype / device when no grid gradient is requested
        shape_carrier = dy.new_empty(x_shape)
        dx, _ = torch.ops.aten.grid_sampler_2d_backward(dy.contiguous(), shape_carrier, grid, 0, 0, False, [True, False])
        return dx

    @staticmethod
    def backward(ctx, ddx):
        grid, = ctx.saved_tensors
        ddy = _Sample.apply(ddx, grid) if ctx.needs_input_grad[0] else None
        return ddy, None, None


#----------------------------------------------------------------------------
# The affine special case on this package's own kernels (ADA's geometric step, training/augment.py).

def affine_sample(x, theta, out_hw):
    """``grid_sample(x, affine_grid(theta, [N, C, *out_hw], align_corners=False))`` (bilinear, zero padding) without the
    grid tensor, differentiable to any order in ``x``: forward ``pasta_affine_sample``; gradient ``pasta_affine_sample_adjoint``,
    a gather over the output lattice points whose footprint covers an input pixel (no atomics); the gradient of the
    gradient is the forward again."""
    return _AffineSample.apply(x, theta, (int(out_hw[0]), int(out_hw[1])))

def _affine_launch(name, src, theta, dst_shape, in_hw, out_hw):
    from . import _native
    _native.require_gpu(src, name)
    if src.dtype != torch.float32 or theta.dtype != torch.float32:
        raise RuntimeError(f'{name}: float32 only')
    n, c = src.shape[0], src.shape[1]
    if theta.shape != (n, 2, 3):
        raise RuntimeError(f'{name}: theta must be [{n}, 2, 3], got {tuple(theta.shape)}')
    src, theta = src.contiguous(), theta.contiguous()
    dst = torch.empty(dst_shape, dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        st = getattr(_native.lib(), name)(_native.ptr(src), _native.ptr(theta), _native.ptr(dst), n, c, in_hw[0], in_hw[1], out_hw[0], out_hw[1],
                                          _native.stream())
    _native.check(st)
    return dst

class _AffineSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, theta, out_hw):
        assert x.ndim == 4
        ctx.save_for_backward(theta)
        ctx.in_hw, ctx.out_hw = (x.shape[2], x.shape[3]), out_hw
        return _affine_launch('pasta_affine_sample', x, theta, [x.shape[0], x.shape[1], *out_hw], ctx.in_hw, out_hw)

    @staticmethod
    def backward(ctx, dy):
        theta, = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise NotImplementedError('affine_sample: no gradient with respect to theta')
        dx = _AffineSampleAdjoint.apply(dy, theta, ctx.in_hw) if ctx.needs_input_grad[0] else None
        return dx, None, None

class _AffineSampleAdjoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, theta, in_hw):
        ctx.save_for_backward(theta)
        ctx.in_hw, ctx.out_hw = in_hw, (dy.shape[2], dy.shape[3])
        return _affine_launch('pasta_affine_sample_adjoint', dy, theta, [dy.shape[0], dy.shape[1], *in_hw], in_hw, ctx.out_hw)

    @staticmethod
    def backward(ctx, ddx):
        theta, = ctx.saved_tensors
        ddy = _AffineSample.apply(ddx, theta, ctx.out_hw) if ctx.needs_input_grad[0] else None
        return ddy, None, None
