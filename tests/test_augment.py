"""ADA augmentation parity: training/augment.py's AugmentPipe.

Fixtures: tests/golden/augment.npz, written by oracle/make_golden_augment.py from the reference's own AugmentPipe run on
the CPU with its random draws recorded.  CPU: the oracle restatement (oracle/ref_augment.py) replays the recorded draws
against the reference's outputs and gradients.  GPU: this package's AugmentPipe (HIP parameter kernel, HIP upfirdn2d,
HIP colour transform, through the C ABI) replays the same draws."""

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import ref_augment as RA
from oracle.make_golden_augment import CASES

IDS = [c['name'] for c in CASES]
TOL_ORACLE = 2e-5       # same fp32 CPU ops in a different arrangement
TOL_HIP = 2e-4          # device sin/cos/exp2/erfinv differ from the CPU's in the last ulp; a parameter that moves by 1e-7
                        # relative shifts sampling positions by <= 1e-4 pixel on these image sizes


def _case_tensors(g, case):
    k = case['name']
    t = {name: torch.from_numpy(g[f'{k}.{name}']) for name in ('x', 'probe', 'u', 'z', 'y', 'dx')}
    t['field'] = torch.from_numpy(g[f'{k}.field']) if f'{k}.field' in g else None
    return t


@pytest.mark.parametrize('idx', range(len(CASES)), ids=IDS)
def test_oracle_matches_reference(idx):
    case, g = CASES[idx], load_golden('augment.npz')
    t = _case_tensors(g, case)
    x = t['x'].clone().requires_grad_(True)
    y = RA.augment(x, t['u'], t['z'], case['cfg'], case['p'], noise_field=t['field'], debug_percentile=case.get('debug_percentile'))
    assert rel_err(y, t['y']) < TOL_ORACLE
    dx, = torch.autograd.grad((y * t['probe']).sum(), x)
    assert rel_err(dx, t['dx']) < TOL_ORACLE


def test_filter_banks_match_reference():
    from training import augment
    g = load_golden('augment.npz')
    assert np.abs(RA.filter_bank() - g['Hz_fbank']).max() < 1e-6
    pipe = augment.AugmentPipe()                       # construction needs no GPU
    assert np.abs(pipe.Hz_fbank.numpy() - g['Hz_fbank']).max() < 1e-6
    assert np.abs(pipe.Hz_geom.numpy() - g['Hz_geom']).max() < 1e-7
    assert sorted(dict(pipe.named_buffers())) == ['Hz_fbank', 'Hz_geom', 'p']


def test_draw_layout_is_shared():
    """The product's column layout, the oracle's, and the kernel's enum (csrc/augment.hip) must agree."""
    from training import augment
    from torch_utils import custom_ops
    assert augment.DRAWS_U == RA.U_COLS and augment.DRAWS_Z == RA.Z_COLS
    cfg_fields = [n for n, _ in custom_ops.AdaConfig._fields_]
    assert all(f in RA.DEFAULTS for f in cfg_fields)
    import os, re
    src = open(os.path.join(os.path.dirname(augment.__file__), '..', 'csrc', 'augment.hip')).read()
    enum_u = re.search(r'enum \{\s*(U_XFLIP_I.*?)U_COLS_MIN', src, re.S).group(1)
    names = [s.strip() for s in enum_u.replace('\n', ' ').split(',') if s.strip()]
    assert len(names) == RA.U_COLS.index('imgfilter.on.0')         # the kernel consumes the geometry + colour columns
    enum_z = re.search(r'enum \{\s*(Z_SCALE.*?)Z_COLS_MIN', src, re.S).group(1)
    assert len([s for s in enum_z.split(',') if s.strip()]) == RA.Z_COLS.index('imgfilter.0')


def test_constructor_signature_matches_reference():
    import inspect
    from training import augment
    params = inspect.signature(augment.AugmentPipe.__mro__[1].__init__).parameters     # [0] is the persistence wrapper
    for name, default in RA.DEFAULTS.items():
        assert params[name].default == default
    pipe = augment.AugmentPipe(xflip=1, imgfilter_bands=[1, 0, 0, 1])
    assert pipe.init_kwargs == dict(xflip=1, imgfilter_bands=[1, 0, 0, 1])
    assert pipe.xflip == 1.0 and pipe.imgfilter_bands == [1, 0, 0, 1] and float(pipe.p) == 1.0


def test_no_cpu_path():
    from training import augment
    pipe = augment.AugmentPipe(xflip=1)
    with pytest.raises(RuntimeError, match='GPU'):
        pipe(torch.zeros([2, 3, 8, 8]))


# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize('idx', range(len(CASES)), ids=IDS)
def test_hip_pipeline_matches_reference(idx):
    from training import augment
    case, g = CASES[idx], load_golden('augment.npz')
    t = _case_tensors(g, case)
    pipe = augment.AugmentPipe(**case['cfg']).cuda()
    pipe.p.copy_(torch.as_tensor(case['p']))
    draws = dict(u=t['u'].cuda(), z=t['z'].cuda())
    if t['field'] is not None:
        draws['noise_field'] = t['field'].cuda()
    x = t['x'].cuda().requires_grad_(True)
    y = pipe(x, debug_percentile=case.get('debug_percentile'), draws=draws)
    assert rel_err(y, t['y']) < TOL_HIP, rel_err(y, t['y'])
    dx, = torch.autograd.grad((y * t['probe'].cuda()).sum(), x)
    assert rel_err(dx, t['dx']) < TOL_HIP, rel_err(dx, t['dx'])


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 7, 300])
def test_parameter_kernel_matches_oracle(n):
    """pasta_ada_matrices against the oracle's matrix construction on fresh draws (n = 300: more samples than threads)."""
    from training import augment
    cfg = dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1)
    pipe = augment.AugmentPipe(**cfg).cuda()
    pipe.p.fill_(0.7)
    gen = torch.Generator().manual_seed(5 + n)
    u = torch.rand([n, len(RA.U_COLS)], generator=gen)
    z = torch.randn([n, len(RA.Z_COLS)], generator=gen)
    W, H = 48, 40
    G, C, margins = pipe.matrices(dict(u=u.cuda(), z=z.cuda()), W, H, 3)
    full = {**RA.DEFAULTS, **cfg}
    Gr = RA.geometry_matrix(u, z, full, torch.tensor(0.7), W, H)
    Cr = RA.color_matrix(u, z, full, torch.tensor(0.7), 3)
    assert rel_err(G, Gr) < 1e-5 and rel_err(C, Cr) < 1e-5
    corners = torch.tensor([[-(W - 1) / 2, -(H - 1) / 2, 1], [(W - 1) / 2, -(H - 1) / 2, 1], [(W - 1) / 2, (H - 1) / 2, 1], [-(W - 1) / 2, (H - 1) / 2, 1]])
    moved = Gr @ corners.t()
    xs, ys = moved[:, 0].flatten(), moved[:, 1].flatten()
    reach = torch.stack([(-xs).max(), (-ys).max(), xs.max(), ys.max()]) + torch.tensor([6 - (W - 1) / 2, 6 - (H - 1) / 2] * 2)
    want = torch.minimum(reach.clamp(min=0), torch.tensor([W - 1.0, H - 1.0] * 2)).ceil().int()
    assert (margins.cpu() - want).abs().max() <= 1            # a margin within 1e-6 of an integer may round either way
    assert margins.dtype == torch.int32


@pytest.mark.gpu
def test_color_transform_gradients():
    """The colour transform is linear in the image: gradient = transposed matrix, second derivative = the matrix again."""
    from training.augment import _ColorAffine
    gen = torch.Generator().manual_seed(3)
    x = torch.randn([3, 3, 9, 7], generator=gen).cuda().requires_grad_(True)     # 63 pixels: the scalar (non-float4) kernel
    C = torch.randn([3, 4, 4], generator=gen).cuda()
    y = _ColorAffine.apply(x, C, 0)
    ref = (C[:, :3, :3] @ x.reshape(3, 3, -1) + C[:, :3, 3:]).reshape_as(x)
    assert rel_err(y, ref) < 1e-6
    d = torch.randn_like(x).requires_grad_(True)
    dx, = torch.autograd.grad(y, x, d, create_graph=True)
    assert rel_err(dx, (C[:, :3, :3].transpose(1, 2) @ d.reshape(3, 3, -1)).reshape_as(x)) < 1e-6
    e = torch.randn_like(x)
    dd, = torch.autograd.grad(dx, d, e)
    assert rel_err(dd, (C[:, :3, :3] @ e.reshape(3, 3, -1)).reshape_as(x)) < 1e-6


@pytest.mark.gpu
def test_full_size_batch_and_r1_double_backward():
    """48 images of 256x256 (one merged discriminator batch of config 2) through 'bgc' with the pipeline's own draws, and
    the R1-style second derivative through the augmentation (loss_wo_flow_fullbody.py:229-247 differentiates D(aug(real))
    twice)."""
    from training import augment
    cfg = dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1)
    pipe = augment.AugmentPipe(**cfg).cuda()
    pipe.p.fill_(0.6)
    torch.manual_seed(0)
    x = (torch.rand([48, 3, 256, 256], device='cuda') * 2 - 1).requires_grad_(True)
    y = pipe(x)
    assert y.shape == x.shape and bool(torch.isfinite(y).all())
    w = torch.randn_like(y)
    g, = torch.autograd.grad((y * w).sum().square(), x, create_graph=True)       # quadratic, so the second derivative is not zero
    g.square().sum().backward()
    assert x.grad is not None and bool(torch.isfinite(x.grad).all()) and float(x.grad.abs().max()) > 0
    # p = 0: every transform is the identity; the resampling chain reproduces the image up to its filters' pass-band error
    pipe.p.fill_(0.0)
    y0 = pipe(x.detach())
    assert rel_err(y0, x.detach()) < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize('hw', [(524, 524), (37, 41), (8, 6)])
def test_sampling_grid_kernel_matches_affine_grid(hw):
    from torch_utils.ops import _native
    gen = torch.Generator().manual_seed(4)
    theta = (torch.randn([5, 2, 3], generator=gen) * 0.7).cuda()
    H, W = hw
    grid = torch.empty([5, H, W, 2], device='cuda')
    _native.check(_native.lib().pasta_ada_grid(_native.ptr(theta), 5, H, W, _native.ptr(grid), _native.stream()))
    ref = torch.nn.functional.affine_grid(theta.double(), [5, 3, H, W], align_corners=False)
    assert rel_err(grid, ref) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('case', [dict(n=3, c=3, ih=40, iw=52, oh=36, ow=44), dict(n=2, c=5, ih=17, iw=9, oh=23, ow=31), dict(n=1, c=1, ih=64, iw=64, oh=8, ow=8)])
def test_affine_sample_matches_grid_sample(case):
    """affine_sample == grid_sample(affine_grid(theta)): values, the gather adjoint against autograd's scatter, and the second
    derivative; maps include rotation, anisotropic scale, a strong zoom-out (many samples per input pixel) and a zoom-in."""
    from torch_utils.ops import grid_sample_gradfix as gs
    n, c = case['n'], case['c']
    gen = torch.Generator().manual_seed(8)
    x = torch.randn([n, c, case['ih'], case['iw']], generator=gen)
    maps = torch.tensor([[[0.9, 0.35, 0.05], [-0.3, 1.1, -0.1]], [[2.6, 0.4, 0.2], [0.1, 2.2, 0.0]], [[0.3, -0.1, 0.4], [0.05, 0.25, -0.3]]])
    theta = maps[torch.arange(n) % 3]
    dy = torch.randn([n, c, case['oh'], case['ow']], generator=gen)
    x64 = x.double().requires_grad_(True)
    grid = torch.nn.functional.affine_grid(theta.double(), [n, c, case['oh'], case['ow']], align_corners=False)
    yr = torch.nn.functional.grid_sample(x64, grid, mode='bilinear', padding_mode='zeros', align_corners=False)
    gr, = torch.autograd.grad(yr, x64, dy.double())
    xg = x.cuda().requires_grad_(True)
    y = gs.affine_sample(xg, theta.cuda(), (case['oh'], case['ow']))
    assert rel_err(y, yr) < 1e-5
    dyg = dy.cuda().requires_grad_(True)
    g, = torch.autograd.grad(y, xg, dyg, create_graph=True)
    assert rel_err(g, gr) < 1e-5
    e = torch.randn(x.shape, generator=gen)
    dd, = torch.autograd.grad(g, dyg, e.cuda())            # d/d(dy) of S^T dy contracted with e = S e
    er = torch.nn.functional.grid_sample(e.double(), grid, mode='bilinear', padding_mode='zeros', align_corners=False)
    assert rel_err(dd, er) < 1e-5
    # the adjoint identity <S x, dy> = <x, S^T dy> in the kernels' own arithmetic
    lhs, rhs = float((y.detach().double() * dy.cuda().double()).sum()), float((xg.detach().double() * g.detach().double()).sum())
    assert abs(lhs - rhs) < 1e-5 * max(abs(lhs), 1.0)
