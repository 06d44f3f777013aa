"""Per-shape timing of the native convolution family (forward, input gradient, weight gradient) on the layer
shapes of the 256x256 'fashion' model at batch 16; optionally next to MIOpen (torch.nn.functional.conv2d) as a
same-hardware reference.  Usage: python tools/bench_conv.py [--miopen] [--reps 10] [--only fwd,dgrad,wgrad | down] [--match text]"""

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan_amd'))
import torch
from torch_utils.ops import conv2d_gradfix as cg

SHAPES = [  # name, N, Cin, H, Cout, k, stride, transposed
    ('spade 128->128 3x3 @128', 16, 128, 128, 128, 3, 1, False),
    ('spade 256->128 3x3 @128', 16, 256, 128, 128, 3, 1, False),
    ('b256 64->64 3x3 @256', 16, 64, 256, 64, 3, 1, False),
    ('down 64->128 3x3 s2 @257 p0', 16, 64, 257, 128, 3, 2, False, 0),      # discriminator: blur to 257, then stride 2 without padding
    ('enc 64->128 3x3 s2 @256 p1', 16, 64, 256, 128, 3, 2, False),
    ('down 128->256 3x3 s2 @129 p0', 16, 128, 129, 256, 3, 2, False, 0),
    ('down 256->512 3x3 s2 @65 p0', 16, 256, 65, 512, 3, 2, False, 0),
    ('down 64->128 3x3 s2 @257 p0 x48', 48, 64, 257, 128, 3, 2, False, 0),
    ('b64 256->256 3x3 @64', 16, 256, 64, 256, 3, 1, False),
    ('b32 512->512 3x3 @32', 16, 512, 32, 512, 3, 1, False),
    ('b16 512->512 3x3 @16', 16, 512, 16, 512, 3, 1, False),
    ('b8 512->512 3x3 @8', 16, 512, 8, 512, 3, 1, False),
    ('b4 512->512 3x3 @4', 16, 512, 4, 512, 3, 1, False),
    ('up 128->64 3x3 T2 @128', 16, 128, 128, 64, 3, 2, True),
    ('up 512->512 3x3 T2 @16', 16, 512, 16, 512, 3, 2, True),
    ('up 256->128 3x3 T2 @64', 16, 256, 64, 128, 3, 2, True),
    ('up 512->256 3x3 T2 @32', 16, 512, 32, 256, 3, 2, True),
    ('up 512->256 3x3 T2 @32 x48', 48, 512, 32, 256, 3, 2, True),
    ('up 512->512 3x3 T2 @16 x48', 48, 512, 16, 512, 3, 2, True),
    ('merge 192->128 1x1 @128', 16, 192, 128, 128, 1, 1, False),
    ('merge 128->64 1x1 @256', 16, 128, 256, 64, 1, 1, False),
    ('skip 64->64 1x1 @256', 16, 64, 256, 64, 1, 1, False),
    ('skip 128->128 1x1 @128', 16, 128, 128, 128, 1, 1, False),
    ('merge 320->256 1x1 @64', 16, 320, 64, 256, 1, 1, False),
    ('skip 64->128 1x1 @128', 16, 64, 128, 128, 1, 1, False),
    ('torgb 64->3 1x1 @256', 16, 64, 256, 3, 1, 1, False),
    ('stem 3->64 7x7 @256', 16, 3, 256, 64, 7, 1, False),
    ('stem 3->64 3x3 @256', 16, 3, 256, 64, 3, 1, False),
    ('fromrgb 3->64 1x1 @256 x48', 48, 3, 256, 64, 1, 1, False),
    ('stem 6->64 1x1 @256', 16, 6, 256, 64, 1, 1, False),
]


def timeit(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--miopen', action='store_true')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--only', default='fwd,dgrad,wgrad')
    ap.add_argument('--match', default='')
    args = ap.parse_args()
    only = args.only.split(',')
    dev = torch.device('cuda')
    print(f"{'shape':28s} {'pass':6s} {'ms':>8s} {'TFLOP/s':>8s}" + (f" {'miopen ms':>10s} {'TF/s':>7s}" if args.miopen else ''))
    for name, n, ci, h, co, k, st, tr, *rest in SHAPES:
        if (args.match and args.match not in name) or only == ['down'] or only == ['s1pieces']:
            continue
        pad = rest[0] if rest else (k // 2 if not tr else 0)
        x = torch.randn([n, ci, h, h], device=dev)
        w = torch.randn([ci, co, k, k] if tr else [co, ci, k, k], device=dev) * 0.05
        cfg = cg._Cfg((tr, st, pad, pad, 0, 0, 1))
        y = cg._launch_conv(x, w, cfg)
        dy = torch.randn_like(y)
        flops = 2.0 * n * ci * co * k * k * ((h * h) if tr else (y.shape[2] * y.shape[3]))
        gcfg = cg._grad_cfg(cfg, x.shape[2:], y.shape[2:], k, k)
        runs = dict(fwd=lambda: cg._launch_conv(x, w, cfg), dgrad=lambda: cg._launch_conv(dy, w, gcfg),
                    wgrad=lambda: cg._launch_wgrad(x, dy, cfg, tuple(w.shape)))
        F = torch.nn.functional
        if tr:
            mi = dict(fwd=lambda: F.conv_transpose2d(x, w, stride=st, padding=pad),
                      dgrad=lambda: F.conv2d(dy, w, stride=st, padding=pad),
                      wgrad=lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [pad, pad], [1, 1], True, [0, 0], 1, [False, True, False]))
        else:
            mi = dict(fwd=lambda: F.conv2d(x, w, stride=st, padding=pad),
                      dgrad=lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]),
                      wgrad=lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]))
        for ps in only:
            ms = timeit(runs[ps], args.reps)
            line = f'{name:28s} {ps:6s} {ms:8.3f} {flops / ms / 1e9:8.1f}'
            if args.miopen:
                try:
                    m2 = timeit(mi[ps], args.reps)
                    line += f' {m2:10.3f} {flops / m2 / 1e9:7.1f}'
                except Exception as ex:  # noqa: BLE001
                    line += f'  miopen failed: {type(ex).__name__}'
            print(line, flush=True)

    if 'down' in only or args.only == 'fwd,dgrad,wgrad':
        down_path(args)
    if 's1pieces' in only or args.only == 'fwd,dgrad,wgrad':
        s1_pieces(args)


DOWN = [  # the down path of conv2d_resample (blur to 2 k + 1, then 3x3 stride 2 without padding): name, N, Cin, H (input), Cout
    ('down 64->128 @256', 16, 64, 256, 128), ('down 64->128 @256 x48', 48, 64, 256, 128), ('down 128->256 @128', 16, 128, 128, 256),
    ('down 128->256 @128 x48', 48, 128, 128, 256), ('down 256->512 @64', 16, 256, 64, 512), ('down 256->512 @64 x48', 48, 256, 64, 512)]


def down_path(args):
    """blur / forward convolution / weight gradient of a down layer, with the blurred tensor as fp32 (PASTA_PIECES=0's path) and as the
    producer-written operand pieces (round 5): per-stage times from events, TFLOP/s of the convolutions, GB/s of the blur on the bytes it moves."""
    from torch_utils.ops import upfirdn2d
    dev = torch.device('cuda')
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).to(dev)
    cfg = cg._Cfg((False, 2, 0, 0, 0, 0, 1))
    print(f"{'down path':28s} {'stage':14s} {'ms':>8s} {'TFLOP/s | GB/s':>14s}")
    for name, n, ci, h, co in DOWN:
        if args.match and args.match not in name:
            continue
        x = torch.randn([n, ci, h, h], device=dev)
        w = torch.randn([co, ci, 3, 3], device=dev) * 0.05
        parts = cg.tensor_amax(x)
        xb = upfirdn2d.upfirdn2d(x, f, padding=[2, 2, 2, 2])
        pieces, bound, shape = cg.blur_pieces(x, f, (2, 2, 2, 2), x_amax=parts)
        y = cg._launch_conv(xb, w, cfg)
        dy = torch.randn_like(y)
        flops = 2.0 * n * ci * co * 9 * y.shape[2] * y.shape[3]
        nbytes = (x.numel() + xb.numel()) * 4
        stages = [('blur fp32', lambda: upfirdn2d.upfirdn2d(x, f, padding=[2, 2, 2, 2]), None), ('blur pieces', lambda: cg.blur_pieces(x, f, (2, 2, 2, 2), x_amax=parts), None),
                  ('fwd fp32', lambda: cg._launch_conv(xb, w, cfg), flops), ('fwd pieces', lambda: cg._launch_conv(pieces, w, cfg, pieces=(bound, shape)), flops),
                  ('wgrad fp32', lambda: cg._launch_wgrad(xb, dy, cfg, tuple(w.shape)), flops),
                  ('wgrad pieces', lambda: cg._launch_wgrad_pieces(pieces, dy, cfg, tuple(w.shape), (bound, shape)), flops)]
        for tag, fn, fl in stages:
            ms = timeit(fn, args.reps)
            print(f'{name:28s} {tag:14s} {ms:8.3f} {(fl / ms / 1e9) if fl else (nbytes / ms / 1e6):14.1f}', flush=True)


def s1_pieces(args):
    """The eight-wave 3x3 stride-1 tile kernel on fp32 x (its maxima attached: no scan in either line) and on the same operand as
    PASTA_LAYOUT_PIECES16 (pasta_pieces_pack: the same pieces, bit for bit): forward and input gradient."""
    from torch_utils.ops import _native
    dev = torch.device('cuda')
    print(f"{'3x3 stride 1':28s} {'operand':14s} {'ms':>8s} {'TFLOP/s':>8s}  equal")
    for name, n, ci, h, co, k, st, tr, *rest in SHAPES:
        if k != 3 or st != 1 or co < 128 or h < 8 or (args.match and args.match not in name):
            continue
        x = torch.randn([n, ci, h, h], device=dev)
        w = torch.randn([co, ci, 3, 3], device=dev) * 0.05
        cfg = cg._Cfg((False, 1, 1, 1, 0, 0, 1))
        _native.amax_attach(x, cg.tensor_amax(x))
        pieces, bound, shape = cg.pieces_pack(x)
        y0 = cg._launch_conv(x, w, cfg)
        try:
            y1 = cg._launch_conv(pieces, w, cfg, pieces=(bound, shape))
        except RuntimeError:        # not the eight-wave tile kernel's launch
            continue
        flops = 2.0 * n * ci * co * 9 * h * h
        for tag, fn in (('fp32', lambda: cg._launch_conv(x, w, cfg)), ('pieces', lambda: cg._launch_conv(pieces, w, cfg, pieces=(bound, shape)))):
            ms = timeit(fn, args.reps)
            print(f'{name:28s} {tag:14s} {ms:8.3f} {flops / ms / 1e9:8.1f}  {bool(torch.equal(y0, y1))}', flush=True)


if __name__ == '__main__':
    main()
