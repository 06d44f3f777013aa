"""Round-4 hardening of the default arithmetic (PASTA_MATH_F16X3):

* operand scales can no longer go stale behind the version counter: the weights are scaled per output row by their packing
  kernel at every launch, and maxima of leaf tensors are never read back from the Python object;
* ``torch.inference_mode()`` (tensors without a version counter) works;
* heavy-tailed operands as training produces them -- one sample / channel 1e4 .. 1e6 times the rest, weights spanning 1e5 --
  keep fp32-class accuracy PER SAMPLE / PER OUTPUT CHANNEL, not only relative to the tensor's largest value, in the forward, the
  input-gradient and the weight-gradient launch, against fp64.

Reference semantics: torch_utils/ops/conv2d_gradfix.py:35-43 (plain fp32 convolutions, whatever was written to the operands).
"""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _conv64(x, w, **kw):
    return torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), **kw)


def _rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


# ----------------------------------------------------------------------------- stale scales

@pytest.mark.parametrize('factor', [8.0, 64.0, 1.0 / 4096])
def test_weight_written_through_data_between_two_forwards(factor):
    """``w.data.mul_()`` / ``w.data.copy_()`` bump no version counter.  Until round 3 the weight's |max| was cached on the Python
    object and the second forward overflowed fp16 (factor >= 4) without an error.  Now nothing about the weight is cached."""
    from training import networks
    g = torch.Generator().manual_seed(3)
    layer = networks.Conv2dLayer(64, 128, kernel_size=3, activation='lrelu').cuda()
    x = torch.randn([8, 64, 48, 48], generator=g).cuda()
    def ref(w):                 # the layer: bias_act(conv(x, w * weight_gain) + b, lrelu) * sqrt(2), in fp64
        z = _conv64(x, w * layer.weight_gain, padding=1) + layer.bias.double().cpu().reshape(1, -1, 1, 1)
        return torch.nn.functional.leaky_relu(z, 0.2) * (2 ** 0.5)
    w0 = layer.weight.detach().clone()
    with torch.no_grad():
        y1 = layer(x)
        layer.weight.data.mul_(factor)                        # no version bump
        y2 = layer(x)
        fresh = torch.randn(layer.weight.shape, generator=g).cuda() * 5
        layer.weight.data.copy_(fresh)
        y3 = layer(x)
    assert torch.isfinite(y2).all() and torch.isfinite(y3).all()
    assert _rel(y1, ref(w0)) < 1e-5 and _rel(y2, ref(w0 * factor)) < 1e-5 and _rel(y3, ref(fresh)) < 1e-5


def test_optimizer_on_data_and_weight_gradient_follow():
    """An optimiser that works on ``.data`` (the reference's torch 1.7 Adam does): three steps, forward + both gradients against
    fp64 after every step."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(4)
    w = (torch.randn([128, 64, 3, 3], generator=g) / 24).cuda().requires_grad_(True)
    x = torch.randn([4, 64, 64, 64], generator=g).cuda()
    dy = torch.randn([4, 128, 64, 64], generator=g).cuda()
    for step in range(3):
        xg = x.clone().requires_grad_(True)
        y = cg.conv2d(xg, w, padding=1)
        gx, gw = torch.autograd.grad(y, [xg, w], dy)
        x64 = x.double().cpu().requires_grad_(True); w64 = w.detach().double().cpu().requires_grad_(True)
        y64 = torch.nn.functional.conv2d(x64, w64, padding=1)
        rx, rw = torch.autograd.grad(y64, [x64, w64], dy.double().cpu())
        assert _rel(y, y64) < 1e-5 and _rel(gx, rx) < 1e-5 and _rel(gw, rw) < 1e-5, step
        w.data.add_(gw, alpha=-30.0 * (step + 1))            # grows the weights by an order of magnitude per step
        x.data.mul_(50.0)                                     # and the input batch: a leaf, never cached


def test_leaf_input_written_through_data():
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(5)
    w = (torch.randn([64, 32, 3, 3], generator=g) / 17).cuda()
    x = torch.randn([4, 32, 64, 64], generator=g).cuda()
    y1 = cg.conv2d(x, w, padding=1)
    x.data.mul_(1000.0)                                       # no version bump
    v = x._version
    y2 = cg.conv2d(x, w, padding=1)
    assert x._version == v
    assert torch.isfinite(y2).all() and _rel(y2, _conv64(x, w, padding=1)) < 1e-5
    assert _rel(y1 * 1000.0, y2) < 1e-5
    assert getattr(x, '_pasta_amax', None) is None           # maxima of a leaf are not kept on the object


def test_scan_results_are_cached_only_where_no_write_can_bypass_the_key():
    from torch_utils.ops import conv2d_gradfix as cg
    t = torch.randn([4, 8, 16, 16]).cuda()
    p1 = cg.tensor_amax(t)
    assert cg.tensor_amax(t) is not p1                       # a leaf outside a backward pass: scanned at every use
    u = t.clone().requires_grad_(True) * 1.0                 # a recorded result: cached per version
    q1 = cg.tensor_amax(u)
    assert cg.tensor_amax(u) is q1
    with torch.no_grad():
        u.mul_(2)
    q2 = cg.tensor_amax(u)
    assert q2 is not q1 and float(q2.max()) == float(u.abs().max())


def test_scan_attached_inside_a_backward_pass_is_not_read_by_a_later_forward():
    """a leaf image batch whose forward ran a kernel that takes no maxima (3 input channels: the fp32 tile) is scanned by the
    weight gradient INSIDE the backward pass, where scan results are attached to the object.  A ``.data`` write afterwards is invisible to the
    (version, pointer) key: a forward that read the attribute back would scale the operand by stale maxima (inf under fp16 pieces)."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(11)
    x = torch.randn([4, 32, 32, 32], generator=g).cuda()     # a leaf batch (no grad_fn, requires no gradient)
    w = (torch.randn([64, 32, 3, 3], generator=g) / 17).cuda().requires_grad_(True)

    class Probe(torch.autograd.Function):                    # meets the leaf inside a backward pass, as _launch_wgrad does with a saved input
        @staticmethod
        def forward(ctx, w_):
            return w_.sum()
        @staticmethod
        def backward(ctx, gy):
            ctx_parts.append(cg.tensor_amax(x))
            return gy * torch.ones_like(w)
    ctx_parts = []
    Probe.apply(w).backward()
    hit = getattr(x, '_pasta_amax', None)
    assert hit is not None and hit[2] is ctx_parts[0] and len(hit) == 3      # attached there, as a scan result
    x.data.mul_(1000.0)                                      # behind the version counter
    p2 = cg.tensor_amax(x)
    assert p2 is not ctx_parts[0] and float(p2.max()) == float(x.abs().max())   # the forward side scans the leaf again
    y = cg.conv2d(x, w.detach(), padding=1)
    assert torch.isfinite(y).all() and _rel(y, _conv64(x, w.detach(), padding=1)) < 1e-5
    # a producer's row (fourth field) stands for its version whoever reads it
    z = cg.conv2d_bias_act(x, w.detach(), None, padding=1, act='lrelu')
    zh = getattr(z, '_pasta_amax', None)
    if zh is not None:
        assert len(zh) == 4 and cg.tensor_amax(z) is zh[2]


def test_check_finite_debug_flag_names_the_launch():
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(6)
    w = (torch.randn([64, 32, 3, 3], generator=g) / 17).cuda()
    x = torch.randn([4, 32, 64, 64], generator=g).cuda()
    x[1, 3, 5, 7] = float('inf')
    old = cg._CHECK_FINITE
    cg._CHECK_FINITE = True
    try:
        with pytest.raises(RuntimeError, match='non-finite output'):
            cg.conv2d(x, w, padding=1)
    finally:
        cg._CHECK_FINITE = old
    y = cg.conv2d(x, w, padding=1)                            # without the flag the non-finite value stays local (sample 1 only)
    assert torch.isfinite(y[0]).all() and torch.isfinite(y[2:]).all() and not torch.isfinite(y[1]).all()


# ----------------------------------------------------------------------------- inference mode

def test_inference_mode_runs_the_default_arithmetic():
    """Inference tensors raise on ``_version``; round 3 crashed here.  Layers, a modulated layer and a block."""
    from training import networks
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(7)
    conv = networks.Conv2dLayer(32, 64, kernel_size=3, activation='lrelu', down=2).cuda()
    syn = networks.SynthesisLayer(64, 64, w_dim=32, resolution=64, up=2).cuda().eval()
    x = torch.randn([4, 32, 64, 64], generator=g).cuda()
    ws = torch.randn([4, 32], generator=g).cuda()
    with torch.no_grad():
        want = syn(conv(x), ws, noise_mode='const')
    with torch.inference_mode():
        got = syn(conv(x), ws, noise_mode='const')
        assert got.is_inference()
        y = cg.conv2d(got, torch.randn([32, 64, 3, 3], generator=g).cuda() / 24, padding=1)
        assert torch.isfinite(y).all()
    assert _rel(got, want) < 1e-6


# ----------------------------------------------------------------------------- heavy tails

def _heavy_case(seed, tail, which):
    """``which``: 'x' -- the activations carry the tails (one sample ``tail`` times the rest, one channel sqrt(tail) times), the
    gradient is plain; 'dy' -- the gradient carries them (a discriminator's per-sample gradients differ by orders of magnitude
    where the logits saturate); 'both' -- the SAME sample is loud in both (an outlier image)."""
    g = torch.Generator().manual_seed(seed)
    n, ci, co, hw = 8, 128, 128, 32
    x = torch.randn([n, ci, hw, hw], generator=g)
    w = torch.randn([co, ci, 3, 3], generator=g) / 34
    w *= torch.logspace(0, -5, co).reshape(-1, 1, 1, 1)       # output channels spanning 1e5
    w[:, ::7] *= 30                                           # and a few loud input channels inside every row
    dy = torch.randn([n, co, hw, hw], generator=g)
    if which in ('x', 'both'):
        x[3] *= tail
        x[:, 17] *= tail ** 0.5
    if which in ('dy', 'both'):
        dy[3] *= tail
        dy[:, 100] *= tail ** 0.5
    return x, w, dy


@pytest.mark.parametrize('which', ['x', 'dy', 'both'])
@pytest.mark.parametrize('tail', [1e4, 1e6])
@pytest.mark.parametrize('mode', ['f16x3', 'bf16x6', 'f32'])
def test_heavy_tailed_operands_per_sample_accuracy(mode, tail, which):
    """Forward and input gradient: error of every (sample, channel) plane relative to THAT plane's largest value; weight gradient:
    every output channel's slice relative to that slice (it sums over all samples, so the loud sample sets the scale of every
    slice -- as it does in fp32).  Bound 1e-5 for all three fp32-class arithmetics."""
    from torch_utils.ops import conv2d_gradfix as cg
    x, w, dy = _heavy_case(11, tail, which)
    e_y, e_x, e_w = _heavy_errors(mode, x, w, dy)
    print(f'{mode} {which} tail {tail:g}: per-plane forward {e_y:.2e}, input gradient {e_x:.2e}, per-row weight gradient {e_w:.2e}')
    assert e_y < 1e-5 and e_x < 1e-5 and e_w < 1e-5


def _heavy_errors(mode, x, w, dy):
    from torch_utils.ops import conv2d_gradfix as cg
    x64 = x.double().requires_grad_(True); w64 = w.double().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, w64, padding=1)
    rx, rw = torch.autograd.grad(y64, [x64, w64], dy.double())
    old = cg.conv_math
    cg.conv_math = mode
    try:
        xc = x.cuda().requires_grad_(True); wc = w.cuda().requires_grad_(True)
        y = cg.conv2d(xc, wc, padding=1)
        gx, gw = torch.autograd.grad(y, [xc, wc], dy.cuda())
    finally:
        cg.conv_math = old
    def per_plane(a, b):            # max over (n, c) planes of  max|a - b| / max|b|  within the plane
        a = a.double().cpu(); b = b.detach()
        num = (a - b).abs().amax(dim=(2, 3)); den = b.abs().amax(dim=(2, 3))
        return float((num / den).max())
    dw = gw.double().cpu()
    return (per_plane(y.detach(), y64), per_plane(gx, rx),
            float(((dw - rw).abs().amax(dim=(1, 2, 3)) / rw.abs().amax(dim=(1, 2, 3))).max()))


def test_weight_gradient_range_limit_is_what_the_header_states():
    """The one configuration per-tensor operand scales cannot serve at fp32 accuracy (include/pasta_hip.h, PASTA_MATH_F16X3:
    "both operands of a weight gradient down to 2^-17 of their tensor's"): sample 3 loud in x (1e6) and sample 5 loud in dy (1e6).
    The two contributions to dw are equally large, and each has one operand 1e-6 .. 1e-9 of ITS tensor's largest element, whose low
    fp16 piece is subnormal.  Forward and input gradient are unaffected (activations: 2^-28); the weight gradient degrades
    gracefully -- measured here and bounded, so that a change of the range shows."""
    g = torch.Generator().manual_seed(11)
    n, ci, co, hw = 8, 128, 128, 32
    x = torch.randn([n, ci, hw, hw], generator=g); x[3] *= 1e6
    w = torch.randn([co, ci, 3, 3], generator=g) / 34
    dy = torch.randn([n, co, hw, hw], generator=g); dy[5] *= 1e6
    e_y, e_x, e_w = _heavy_errors('f16x3', x, w, dy)
    print(f'crossed tails 1e6: forward {e_y:.2e}, input gradient {e_x:.2e}, weight gradient {e_w:.2e}')
    assert e_y < 1e-5 and e_x < 1e-5
    assert e_w < 2e-4                # 2^-11 of a term at 2^-20 of its tensor: measured ~1e-5 .. 1e-4, never a collapse


def test_weight_rows_carry_their_own_scale():
    """Output channels whose weights are 1e-9 of the loudest channel's keep full accuracy (one scale per tensor left their low
    pieces in fp16's subnormal range: absolute error 2^-28 of the TENSOR's largest weight per term)."""
    from torch_utils.ops import conv2d_gradfix as cg
    g = torch.Generator().manual_seed(13)
    x = torch.randn([4, 64, 64, 64], generator=g)
    w = torch.randn([128, 64, 3, 3], generator=g) / 24
    w[64:] *= 1e-9
    w[5] *= 1e6
    ref = _conv64(x, w, padding=1)
    y = cg.conv2d(x.cuda(), w.cuda(), padding=1).double().cpu()
    per_row = ((y - ref).abs().amax(dim=(0, 2, 3)) / ref.abs().amax(dim=(0, 2, 3)))
    assert float(per_row.max()) < 3e-6, per_row
    # and the input gradient (rows of that launch = input channels: every row mixes loud and quiet output channels, as fp32 does)
    dy = torch.randn([4, 128, 64, 64], generator=g)
    xg = x.cuda().requires_grad_(True)
    gx, = torch.autograd.grad(cg.conv2d(xg, w.cuda(), padding=1), xg, dy.cuda())
    x64 = x.double().requires_grad_(True)
    rx, = torch.autograd.grad(torch.nn.functional.conv2d(x64, w.double(), padding=1), x64, dy.double())
    assert _rel(gx, rx) < 3e-6


def test_per_sample_modulated_weights_run_three_products():
    """pasta_conv2d_modulated under the default arithmetic (round 3 fell back to six products: the per-sample |max| was unknown).
    Samples whose styles differ by 1e4 keep per-sample accuracy: the scale is per (sample, output channel) row."""
    import ctypes
    from torch_utils.ops import conv2d_gradfix as cg
    from torch_utils import custom_ops
    g = torch.Generator().manual_seed(14)
    n, ci, co, hw = 8, 64, 64, 128                  # 16384 pixels per group: the matrix-core tile
    x = torch.randn([n, ci, hw, hw], generator=g)
    w = torch.randn([co, ci, 3, 3], generator=g) / 24
    s = torch.randn([n, ci], generator=g)
    s[2] *= 1e4; s[6] *= 1e-3
    d = (((w.double()[None] * s.double()[:, None, :, None, None]) ** 2).sum(dim=(2, 3, 4)) + 1e-8).rsqrt().float()
    with torch.no_grad():
        y = cg.modulated_conv2d_forward(x.cuda(), w.cuda(), s.cuda(), d.cuda(), padding=1, per_sample=True)
    wmod = (w.double()[None] * s.double()[:, None, :, None, None]) * d.double()[:, :, None, None, None]
    ref = torch.cat([torch.nn.functional.conv2d(x[i:i + 1].double(), wmod[i], padding=1) for i in range(n)])
    per_sample = (y.double().cpu() - ref).abs().amax(dim=(1, 2, 3)) / ref.abs().amax(dim=(1, 2, 3))
    assert float(per_sample.max()) < 1e-5, per_sample
    desc = custom_ops.ConvDesc(N=1, C_in=n * ci, H=hw, W=hw, C_out=n * co, OH=hw, OW=hw, kh=3, kw=3, stride=1, pad_h=1, pad_w=1, groups=n,
                               transposed=0, flip=0, math=0)
    math = ctypes.c_int()
    assert custom_ops.get_plugin().pasta_conv2d_plan(ctypes.byref(desc), 8, None, None, ctypes.byref(math), None, None) == 0
    assert math.value == cg.MATH_CODES['f16x3']
