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
