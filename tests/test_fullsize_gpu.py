"""BASELINE config-2 sizes (batch 16, 'fashion' widths) through size-independent properties: adjointness of every
linear operator with its hand-written gradient kernels, linearity, and statistics identities. No oracle needed, so the
production shapes, tile paths and split-K / ksplit plans are the ones exercised."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _rel(a, b):
    return abs(a - b) / (abs(b) + 1e-30)


CONV_SHAPES = [  # x shape, w shape, kwargs -- one of each kernel family at full size
    ([16, 128, 128, 128], [128, 128, 3, 3], dict(padding=1)),                      # SPADE 3x3: 128x128 tile, wgrad 3x3 pipelined
    ([16, 256, 128, 128], [128, 256, 3, 3], dict(padding=1)),
    ([16, 64, 256, 256], [64, 64, 3, 3], dict(padding=1)),                         # 64x256 tile
    ([16, 64, 256, 256], [128, 64, 3, 3], dict(down=2, padding=1)),                # blur + stride-2 conv (16-pixel wgrad chunks)
    ([16, 128, 128, 128], [64, 128, 3, 3], dict(up=2, padding=1, flip_weight=False)),   # transposed conv (4 lattices) + blur
    ([16, 512, 8, 8], [512, 512, 3, 3], dict(padding=1)),                          # split-K forward
    ([16, 512, 4, 4], [512, 512, 3, 3], dict(up=2, padding=1, flip_weight=False)),
    ([16, 192, 128, 128], [128, 192, 1, 1], dict()),                               # merge conv, 1x1 wgrad with 2x2 wave tiles
    ([16, 3, 256, 256], [64, 3, 7, 7], dict(padding=3)),                           # RGB stem: 4-channel K chunks, small-Cin wgrad
    ([16, 64, 256, 256], [3, 64, 1, 1], dict()),                                   # ToRGB
    ([16, 64, 256, 256], [128, 64, 1, 1], dict(down=2)),                           # discriminator skip
    ([16, 128, 256, 256], [64, 128, 1, 1], dict()),                                # round 4: pointwise kernel, 64-row tile (and the 128-row one for its input gradient)
    ([48, 3, 256, 256], [64, 3, 1, 1], dict()),                                    # fromrgb over the stacked discriminator batch: few-channel pointwise weight gradient
    ([16, 128, 129, 129], [256, 128, 3, 3], dict(down=2, padding=1)),              # stride-2 kernel on two tile rows of 64
]


@pytest.mark.parametrize('idx', range(len(CONV_SHAPES)))
def test_conv2d_resample_adjoint_identities_full_size(idx):
    """<dy, A(x, w)> = <A_x^T dy, x> = <A_w^T dy, w> for the bilinear map A = conv2d_resample."""
    from torch_utils.ops import conv2d_resample, upfirdn2d
    xs, ws, kw = CONV_SHAPES[idx]
    g = torch.Generator(device='cuda').manual_seed(idx)
    x = torch.randn(xs, device='cuda', generator=g).requires_grad_(True)
    w = (torch.randn(ws, device='cuda', generator=g) / np.sqrt(ws[1] * ws[2] * ws[3])).requires_grad_(True)
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
    y = conv2d_resample.conv2d_resample(x, w, f=f, **kw)
    dy = torch.randn(y.shape, device='cuda', generator=g)
    dx, dw = torch.autograd.grad(y, [x, w], dy)
    lhs = _dot(dy, y)
    assert torch.isfinite(y).all()
    assert _rel(_dot(dx, x), lhs) < 2e-4, (_dot(dx, x), lhs)
    assert _rel(_dot(dw, w), lhs) < 2e-4, (_dot(dw, w), lhs)
    # linearity in x (one more forward): A(2x - 3x') = 2A(x) - 3A(x')
    with torch.no_grad():
        x2 = torch.randn(xs, device='cuda', generator=g)
        y2 = conv2d_resample.conv2d_resample(x2, w, f=f, **kw)
        y3 = conv2d_resample.conv2d_resample(2 * x - 3 * x2, w, f=f, **kw)
        err = (y3 - (2 * y - 3 * y2)).abs().max() / y3.abs().max()
    assert float(err) < 1e-4


UPF_SHAPES = [
    ([16, 64, 257, 257], dict(padding=[1, 1, 1, 1], gain=4)),
    ([16, 64, 256, 256], dict(padding=[2, 2, 2, 2])),
    ([16, 64, 256, 256], dict(down=2, padding=[1, 1, 1, 1])),
    ([16, 3, 128, 128], dict(up=2, padding=[2, 1, 2, 1], gain=4)),
    ([16, 512, 9, 9], dict(padding=[1, 1, 1, 1], gain=4)),
]


@pytest.mark.parametrize('idx', range(len(UPF_SHAPES)))
def test_upfirdn2d_adjoint_and_dc_gain_full_size(idx):
    from torch_utils.ops import upfirdn2d
    shape, kw = UPF_SHAPES[idx]
    g = torch.Generator(device='cuda').manual_seed(10 + idx)
    x = torch.randn(shape, device='cuda', generator=g).requires_grad_(True)
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
    y = upfirdn2d.upfirdn2d(x, f, **kw)
    dy = torch.randn(y.shape, device='cuda', generator=g)
    dx, = torch.autograd.grad(y, x, dy)
    assert _rel(_dot(dx, x), _dot(dy, y)) < 1e-4
    # a normalised low-pass maps a constant image to gain/(up^2) times the constant away from the borders
    with torch.no_grad():
        c = upfirdn2d.upfirdn2d(torch.ones(shape, device='cuda'), f, **kw)
        up = kw.get('up', 1)
        inner = c[:, :, 4:-4, 4:-4] if c.shape[2] > 12 else c[:, :, 2:-2, 2:-2]
        assert float((inner - kw.get('gain', 1) / up ** 2).abs().max()) < 1e-5


def test_bias_act_and_spade_identities_full_size():
    from torch_utils.ops import bias_act
    from training import networks
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn([16, 64, 256, 256], device='cuda', generator=g).requires_grad_(True)
    b = torch.randn([64], device='cuda', generator=g).requires_grad_(True)
    y = bias_act.bias_act(x, b, act='lrelu', gain=np.sqrt(2), clamp=256)
    # lrelu is positively homogeneous: y == sqrt(2) * max(u, 0.2u)
    u = x.detach() + b.detach().reshape(1, -1, 1, 1)
    assert float((y.detach() - np.sqrt(2) * torch.maximum(u, 0.2 * u)).abs().max()) < 1e-5
    dy = torch.randn(y.shape, device='cuda', generator=g)
    dx, db = torch.autograd.grad(y, [x, b], dy)
    assert _rel(float(db.double().sum()), float(dx.double().sum())) < 1e-5      # db is the plane sum of dx
    # SPADE: with gamma = beta = 0 every plane comes out with zero mean and unit (biased) variance
    xs = torch.randn([16, 128, 128, 128], device='cuda', generator=g) * 3 + 1
    z = torch.zeros_like(xs)
    o = networks.spade_modulate(xs, z, z)
    assert float(o.mean(dim=[2, 3]).abs().max()) < 1e-5
    assert float((o.var(dim=[2, 3], unbiased=False) - 1).abs().max()) < 1e-3


def test_one_training_iteration_batch16():
    """The benchmark's step itself: finite, every trainable parameter that should move does."""
    from training.training_loop_wo_flow_fullbody import TrainingStep, SyntheticFullBodyBatch
    dev = torch.device('cuda', 0)
    step = TrainingStep(dev, batch_size=16, batch_gpu=16)
    data = SyntheticFullBodyBatch(16, dev, seed=0)
    g0 = [p.detach().clone() for p in step.G.parameters()]
    d0 = [p.detach().clone() for p in step.D.parameters()]
    step.run(data)
    torch.cuda.synchronize()
    moved_g = sum(int(not torch.equal(a, b)) for a, b in zip(g0, step.G.parameters()))
    moved_d = sum(int(not torch.equal(a, b)) for a, b in zip(d0, step.D.parameters()))
    assert moved_d == len(d0)
    assert moved_g >= len(g0) - 6            # b4.const and the texture block's unused parsing / conv0-less parameters never get gradients
    assert all(torch.isfinite(p).all() for p in step.G.parameters()) and all(torch.isfinite(p).all() for p in step.D.parameters())


def test_merge_layer_over_two_tensors_full_size():
    """The synthesis blocks' merge at 256 x 256 (networks.py:5698-5700) as ONE pointwise launch over (x, side): equal to the convolution of the
    concatenation, and adjoint to its three gradients."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator(device='cuda').manual_seed(77)
    x = torch.randn([16, 64, 256, 256], device='cuda', generator=g).requires_grad_(True)
    side = (torch.randn([16, 64, 256, 256], device='cuda', generator=g) * 0.1).requires_grad_(True)
    w = (torch.randn([64, 128, 1, 1], device='cuda', generator=g) / np.sqrt(128)).requires_grad_(True)
    b = torch.randn([64], device='cuda', generator=g).requires_grad_(True)
    assert cg.cat1x1_available(x, side, w)
    y = cg.conv2d_cat1x1_bias_act(x, side, w, b, act='linear', clamp=256)
    with torch.no_grad():
        ref = cg.conv2d_bias_act(torch.cat([x, side], dim=1), w, b, act='linear', clamp=256)
    assert float((y - ref).abs().max() / ref.abs().max()) < 2e-6
    dy = torch.randn(y.shape, device='cuda', generator=g)
    dx, ds, dw, db = torch.autograd.grad(y, [x, side, w, b], dy)
    # y is affine in (x, side) jointly with w: <dy, y - bias> = <dx, x> + <ds, side> = <dw, w>   (no clamp is active at these magnitudes)
    assert float(y.abs().max()) < 256
    lhs = _dot(dy, y) - _dot(db, b)
    assert _rel(_dot(dx, x) + _dot(ds, side), lhs) < 2e-4 and _rel(_dot(dw, w), lhs) < 2e-4
