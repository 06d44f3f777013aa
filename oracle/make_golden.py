"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    python oracle/make_golden.py [--ref /root/reference] [--only ops|layers|primitives|models|models_f64|fullwidth|fullwidth_r1|loss|augment|snapshot]

The reference is imported read-only from ``--ref`` (never copied): op level as is, model level
with the two harness accommodations SURVEY.md F1/F4 describe (cwd = reference root,
``torch.version.cuda`` preset so the dead custom-kernel probe in training/networks.py:1206 is
skipped). Outputs are small fp32 vectors (inputs + reference outputs + gradients); the cases
and seeds are listed in ``CASES_*`` below so a reader can see exactly what was pinned.
This script is the only place that touches the reference; tests read the .npz files.
"""

import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float64) * scale).to(dtype)

def to_np(t):
    return t.detach().cpu().numpy()

#----------------------------------------------------------------------------
# Case tables (shared with the tests through the manifest stored in each file).

F4 = [1, 3, 3, 1]
F12 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633,
       0.4910559419267466, 0.787641141030194, 0.3379294217276218, -0.07263752278646252,
       -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148]  # sym6 (augment.py:41)

CASES_UPFIRDN2D = [
    # name, x shape, filter taps (None = identity), setup kwargs, call kwargs
    dict(name='blur_pad2',      shape=[2, 3, 9, 9],   f=F4,  call=dict(padding=[2, 2, 2, 2])),
    dict(name='blur_pad1_g4',   shape=[2, 3, 17, 17], f=F4,  call=dict(padding=[1, 1, 1, 1], gain=4)),
    dict(name='up2_rgb',        shape=[2, 3, 8, 8],   f=F4,  call=dict(up=2, padding=[2, 1, 2, 1], gain=4)),
    dict(name='down2_skip',     shape=[2, 5, 16, 12], f=F4,  call=dict(down=2, padding=[1, 1, 1, 1])),
    dict(name='down2_odd',      shape=[1, 4, 17, 13], f=F4,  call=dict(down=2, padding=[1, 1, 1, 1])),
    dict(name='up2_flip',       shape=[1, 4, 7, 9],   f=[1, 2, 4, 1], call=dict(up=2, padding=[2, 1, 2, 1], flip_filter=True, gain=4)),
    dict(name='blur_flip_asym', shape=[2, 2, 10, 11], f=[1, 2, 4, 1], call=dict(padding=[2, 1, 1, 2], flip_filter=True)),
    dict(name='crop_negpad',    shape=[1, 3, 16, 16], f=F4,  call=dict(padding=[-2, -1, -3, 0], gain=1)),
    dict(name='identity',       shape=[2, 3, 6, 5],   f=None, call=dict(padding=[1, 0, 2, 1])),
    dict(name='sep12_up2',      shape=[1, 3, 20, 18], f=F12, call=dict(up=2, padding=[7, 6, 7, 6], gain=4)),
    dict(name='sep12_down2',    shape=[1, 3, 40, 36], f=F12, call=dict(down=2, padding=[-3, -2, -3, -2], flip_filter=True)),
    dict(name='up3_down2_5tap', shape=[1, 2, 9, 8],   f=[1, 4, 6, 4, 1], call=dict(up=3, down=2, padding=[3, 2, 4, 1], gain=9)),
    dict(name='updown_xy',      shape=[1, 2, 8, 10],  f=F4,  call=dict(up=[2, 1], down=[1, 2], padding=[2, 1, 1, 1], gain=2)),
    dict(name='blur_64',        shape=[1, 2, 65, 65], f=F4,  call=dict(padding=[1, 1, 1, 1], gain=4)),
    dict(name='blur_wide',      shape=[1, 1, 33, 130], f=F4, call=dict(padding=[2, 2, 2, 2])),
    dict(name='up2_wide',       shape=[1, 2, 24, 70], f=F4,  call=dict(up=2, padding=[2, 1, 2, 1], gain=4)),
    dict(name='down2_wide',     shape=[1, 2, 48, 140], f=F4, call=dict(down=2, padding=[1, 1, 1, 1])),
    dict(name='blur3x3',        shape=[1, 2, 30, 70], f=[1, 2, 1], call=dict(padding=[1, 1, 1, 1])),
]

ACTS = ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']
CASES_BIAS_ACT = (
    [dict(name=f'{a}_default', shape=[2, 5, 6, 4], act=a, bias=True, kw=dict()) for a in ACTS] +
    [dict(name=f'{a}_gain_clamp', shape=[3, 4, 5], act=a, bias=True, kw=dict(gain=1.7, clamp=0.6, dim=1)) for a in ACTS] +
    [dict(name='lrelu_alpha_dim2', shape=[2, 3, 7], act='lrelu', bias=True, kw=dict(alpha=0.1, dim=2, gain=0.5)),
     dict(name='lrelu_nobias_clamp256', shape=[2, 8, 8, 8], act='lrelu', bias=False, kw=dict(gain=float(np.sqrt(2)), clamp=256.0)),
     dict(name='linear_fc', shape=[4, 16], act='linear', bias=True, kw=dict()),
     dict(name='relu_nobias_g', shape=[2, 4, 9, 9], act='relu', bias=False, kw=dict(gain=float(np.sqrt(0.5)))),
     dict(name='linear_clamp', shape=[2, 3, 8, 8], act='linear', bias=True, kw=dict(clamp=0.5)),
     dict(name='sigmoid_dim0', shape=[6, 5], act='sigmoid', bias=True, kw=dict(dim=0))])

CASES_CONV2D_RESAMPLE = [
    dict(name='plain3x3',       x=[2, 4, 9, 9],   w=[6, 4, 3, 3], kw=dict(padding=1)),
    dict(name='plain3x3_noflip', x=[2, 4, 9, 9],  w=[6, 4, 3, 3], kw=dict(padding=1, flip_weight=False)),
    dict(name='plain1x1',       x=[2, 5, 8, 8],   w=[3, 5, 1, 1], kw=dict()),
    dict(name='plain7x7',       x=[1, 3, 12, 12], w=[4, 3, 7, 7], kw=dict(padding=3)),
    dict(name='down2_3x3',      x=[2, 4, 16, 16], w=[6, 4, 3, 3], f=F4, kw=dict(down=2, padding=1)),
    dict(name='down2_1x1',      x=[2, 4, 16, 16], w=[6, 4, 1, 1], f=F4, kw=dict(down=2)),
    dict(name='up2_3x3',        x=[2, 4, 8, 8],   w=[6, 4, 3, 3], f=F4, kw=dict(up=2, padding=1, flip_weight=False)),
    dict(name='up2_3x3_flipw',  x=[2, 4, 8, 8],   w=[6, 4, 3, 3], f=F4, kw=dict(up=2, padding=1, flip_weight=True)),
    dict(name='up2_1x1',        x=[2, 4, 8, 8],   w=[3, 4, 1, 1], f=F4, kw=dict(up=2)),
    dict(name='grouped3x3',     x=[1, 8, 9, 9],   w=[6, 4, 3, 3], kw=dict(padding=1, groups=2)),
    dict(name='grouped_up2',    x=[1, 8, 6, 6],   w=[6, 4, 3, 3], f=F4, kw=dict(up=2, padding=1, groups=2, flip_weight=False)),
    dict(name='grouped_down2',  x=[1, 8, 12, 12], w=[6, 4, 3, 3], f=F4, kw=dict(down=2, padding=1, groups=2)),
    dict(name='asym_pad_generic', x=[1, 3, 9, 9], w=[2, 3, 3, 3], kw=dict(padding=[1, 0, 2, 1])),
    dict(name='updown_generic', x=[1, 3, 8, 8],   w=[2, 3, 3, 3], f=F4, kw=dict(up=2, down=2, padding=1)),
    dict(name='odd_channels',   x=[2, 7, 10, 10], w=[5, 7, 3, 3], kw=dict(padding=1)),
]

CASES_MODCONV = [
    dict(name='demod_noise',    x=[2, 4, 8, 8], w=[6, 4, 3, 3], noise='sample', kw=dict(padding=1)),
    dict(name='demod_constnoise', x=[2, 4, 8, 8], w=[6, 4, 3, 3], noise='const', kw=dict(padding=1)),
    dict(name='demod_up2',      x=[2, 4, 8, 8], w=[6, 4, 3, 3], f=F4, noise='sample16', kw=dict(up=2, padding=1, flip_weight=False)),
    dict(name='torgb',          x=[2, 8, 8, 8], w=[3, 8, 1, 1], noise=None, kw=dict(demodulate=False)),
    dict(name='demod_nonoise',  x=[3, 5, 6, 6], w=[4, 5, 3, 3], noise=None, kw=dict(padding=1)),
]

#----------------------------------------------------------------------------

def gen_ops(ref_root):
    sys.path.insert(0, ref_root)
    from torch_utils.ops import upfirdn2d as r_up, bias_act as r_ba, conv2d_resample as r_cr, fma as r_fma

    # upfirdn2d: y and dx for a seeded dy.
    out = {}
    for i, c in enumerate(CASES_UPFIRDN2D):
        x = rnd(c['shape'], 100 + i).requires_grad_(True)
        f = r_up.setup_filter(c['f']) if c['f'] is not None else None
        y = r_up.upfirdn2d(x, f, impl='ref', **c['call'])
        dy = rnd(list(y.shape), 500 + i)
        dx, = torch.autograd.grad(y, x, dy)
        out[c['name'] + '.x'] = to_np(x); out[c['name'] + '.y'] = to_np(y)
        out[c['name'] + '.dy'] = to_np(dy); out[c['name'] + '.dx'] = to_np(dx)
        if f is not None:
            out[c['name'] + '.f'] = to_np(f)
    out['manifest'] = np.array(json.dumps(CASES_UPFIRDN2D))
    np.savez_compressed(os.path.join(GOLDEN, 'ops_upfirdn2d.npz'), **out)

    # setup_filter variants.
    out = {}
    specs = [dict(f=F4), dict(f=F4, gain=4), dict(f=F12), dict(f=[1, 2, 4, 1], flip_filter=True), dict(f=None),
             dict(f=[1, 3, 3, 1], separable=True), dict(f=[[1, 2], [3, 4]], normalize=False, flip_filter=True, gain=2), dict(f=3.0)]
    for i, s in enumerate(specs):
        out[f'f{i}'] = to_np(r_up.setup_filter(**s))
    out['manifest'] = np.array(json.dumps(specs))
    np.savez_compressed(os.path.join(GOLDEN, 'ops_setup_filter.npz'), **out)

    # bias_act: y, dx, db and the second-order terms the R1 penalty exercises.
    out = {}
    for i, c in enumerate(CASES_BIAS_ACT):
        kw = dict(c['kw'])
        dim = kw.get('dim', 1)
        x = rnd(c['shape'], 1000 + i).requires_grad_(True)
        b = rnd([c['shape'][dim]], 2000 + i, 0.5).requires_grad_(True) if c['bias'] else None
        y = r_ba.bias_act(x, b, act=c['act'], impl='ref', **kw)
        dy = rnd(list(y.shape), 3000 + i).requires_grad_(True)
        grads = torch.autograd.grad(y, [x] + ([b] if b is not None else []), dy, create_graph=True)
        dx = grads[0]
        ddx = rnd(list(x.shape), 4000 + i)
        gg = torch.autograd.grad(dx, [dy, x], ddx, allow_unused=True)
        n = c['name']
        out[n + '.x'] = to_np(x); out[n + '.y'] = to_np(y); out[n + '.dy'] = to_np(dy); out[n + '.dx'] = to_np(dx)
        if b is not None:
            out[n + '.b'] = to_np(b); out[n + '.db'] = to_np(grads[1])
        out[n + '.ddx'] = to_np(ddx)
        out[n + '.g_dy'] = to_np(gg[0])
        out[n + '.g_x'] = to_np(gg[1]) if gg[1] is not None else np.zeros(c['shape'], np.float32)
    out['manifest'] = np.array(json.dumps(CASES_BIAS_ACT))
    np.savez_compressed(os.path.join(GOLDEN, 'ops_bias_act.npz'), **out)

    # conv2d_resample: y, dx, dw.
    out = {}
    for i, c in enumerate(CASES_CONV2D_RESAMPLE):
        x = rnd(c['x'], 5000 + i).requires_grad_(True)
        w = rnd(c['w'], 6000 + i, 0.3).requires_grad_(True)
        f = r_up.setup_filter(c['f']) if c.get('f') is not None else None
        y = r_cr.conv2d_resample(x, w, f=f, **c['kw'])
        dy = rnd(list(y.shape), 7000 + i)
        dx, dw = torch.autograd.grad(y, [x, w], dy)
        n = c['name']
        out[n + '.x'] = to_np(x); out[n + '.w'] = to_np(w); out[n + '.y'] = to_np(y)
        out[n + '.dy'] = to_np(dy); out[n + '.dx'] = to_np(dx); out[n + '.dw'] = to_np(dw)
    out['manifest'] = np.array(json.dumps(CASES_CONV2D_RESAMPLE))
    np.savez_compressed(os.path.join(GOLDEN, 'ops_conv2d_resample.npz'), **out)

    # fma with the broadcast pattern of modulated_conv2d (networks.py:77).
    out = {}
    a = rnd([2, 3, 4, 5], 8000).requires_grad_(True)
    b = rnd([2, 3, 1, 1], 8001).requires_grad_(True)
    c = rnd([2, 1, 4, 5], 8002).requires_grad_(True)
    y = r_fma.fma(a, b, c)
    dy = rnd(list(y.shape), 8003)
    da, db, dc = torch.autograd.grad(y, [a, b, c], dy)
    for k, v in dict(a=a, b=b, c=c, y=y, dy=dy, da=da, db=db, dc=dc).items():
        out[k] = to_np(v)
    np.savez_compressed(os.path.join(GOLDEN, 'ops_fma.npz'), **out)
    print('ops fixtures written to', GOLDEN)

#----------------------------------------------------------------------------

def import_reference_networks(ref_root):
    """training.networks of the reference, with the SURVEY F1/F4 accommodations."""
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    os.chdir(ref_root)                    # util_functions.py:11 loads ./human_colormap.mat
    if torch.version.cuda is None:
        torch.version.cuda = '11.0'       # networks.py:1206 parses it; cu110 keeps the dead branch off
    import training.networks as rn
    return rn

def gen_layers(ref_root):
    rn = import_reference_networks(ref_root)
    from torch_utils.ops import upfirdn2d as r_up
    out = {}
    for i, c in enumerate(CASES_MODCONV):
        x = rnd(c['x'], 9000 + i).requires_grad_(True)
        w = rnd(c['w'], 9100 + i, 0.5).requires_grad_(True)
        s = (rnd([c['x'][0], c['x'][1]], 9200 + i, 0.5) + 1).requires_grad_(True)
        f = r_up.setup_filter(c['f']) if c.get('f') is not None else None
        res = c['x'][2] * c['kw'].get('up', 1)
        noise = None
        if c['noise'] in ('sample', 'sample16'):
            noise = rnd([c['x'][0], 1, res, res], 9300 + i, 0.3)
        elif c['noise'] == 'const':
            noise = rnd([res, res], 9300 + i, 0.3)
        n = c['name']
        for fused in (False, True):
            y = rn.modulated_conv2d(x=x, weight=w, styles=s, noise=noise, resample_filter=f, fused_modconv=fused, **c['kw'])
            dy = rnd(list(y.shape), 9400 + i)
            dx, dw, ds = torch.autograd.grad(y, [x, w, s], dy)
            tag = n + ('.fused' if fused else '.plain')
            out[tag + '.y'] = to_np(y); out[tag + '.dx'] = to_np(dx); out[tag + '.dw'] = to_np(dw); out[tag + '.ds'] = to_np(ds)
        out[n + '.x'] = to_np(x); out[n + '.w'] = to_np(w); out[n + '.s'] = to_np(s); out[n + '.dy'] = to_np(dy)
        if noise is not None:
            out[n + '.noise'] = to_np(noise)
    out['manifest'] = np.array(json.dumps(CASES_MODCONV))
    np.savez_compressed(os.path.join(GOLDEN, 'layers_modconv.npz'), **out)
    print('layer fixtures written to', GOLDEN)

#----------------------------------------------------------------------------

def gen_primitives(ref_root):
    """Layer-level cases (oracle/primitive_cases.py) run through the reference's own classes."""
    rn = import_reference_networks(ref_root)
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import primitive_cases as PC
    out = {}
    for idx, case in enumerate(PC.CASES):
        for k, v in PC.run_case(rn, case, idx).items():
            out[case['name'] + '.' + k] = to_np(v)
    np.savez_compressed(os.path.join(GOLDEN, 'layers_primitives.npz'), **out)
    print('primitive fixtures written:', len(PC.CASES), 'cases,', len(out), 'arrays')

#----------------------------------------------------------------------------

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--only', default='all')
    args = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    if args.only in ('all', 'ops'):
        gen_ops(args.ref)
    if args.only in ('all', 'layers'):
        gen_layers(args.ref)
    if args.only in ('all', 'primitives'):
        gen_primitives(args.ref)
    if args.only in ('all', 'models'):
        from make_golden_models import gen_models
        gen_models(args.ref, import_reference_networks)
    if args.only in ('all', 'models_f64'):
        from make_golden_models_f64 import gen_models_f64
        gen_models_f64(args.ref, import_reference_networks)
    if args.only in ('all', 'fullwidth'):
        from make_golden_fullwidth import gen_fullwidth
        gen_fullwidth(args.ref, import_reference_networks)
    if args.only in ('all', 'fullwidth_r1'):
        from make_golden_fullwidth_r1 import gen_fullwidth_r1
        gen_fullwidth_r1(args.ref, import_reference_networks)
    if args.only == 'snapshot':       # own process: it blanks the reference's module-source capture before importing its networks
        from make_golden_snapshot import gen_snapshot
        gen_snapshot(args.ref, import_reference_networks)
    if args.only in ('all', 'loss'):
        from make_golden_loss import gen_loss
        gen_loss(args.ref, import_reference_networks)
    if args.only in ('all', 'augment'):
        from make_golden_augment import gen_augment
        gen_augment(args.ref)
