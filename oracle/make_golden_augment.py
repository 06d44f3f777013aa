"""ADA augmentation golden vectors, build container only.

Runs the REFERENCE'S OWN ``training.augment.AugmentPipe`` on the CPU (imported read-only from /root/reference) with
``torch.rand`` / ``torch.randn`` wrapped so that every number the pipeline draws is recorded, lays the recorded numbers
into the ``u [N,29]`` / ``z [N,12]`` columns of ``oracle/ref_augment.py`` (``DRAW_ORDER``), and writes
tests/golden/augment.npz: per case the input images, the draws, the reference's output and the gradient of a seeded probe
with respect to the images.  ``debug_percentile`` cases need no draws.  Also stored: the reference's ``Hz_fbank`` and
``Hz_geom`` buffers.

Called by ``oracle/make_golden.py --only augment``."""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_augment as RA  # noqa: E402

BGC = dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1)
CASES = [
    dict(name='bgc_p1',        cfg=BGC, p=1.0, shape=[4, 3, 40, 48], seed=11),
    dict(name='bgc_p05',       cfg=BGC, p=0.5, shape=[6, 3, 32, 32], seed=12),
    dict(name='bgc_p02_big',   cfg=BGC, p=0.2, shape=[3, 3, 64, 64], seed=13),
    dict(name='blit',          cfg=dict(xflip=1, rotate90=1, xint=1), p=1.0, shape=[5, 3, 24, 32], seed=14),
    dict(name='geom',          cfg=dict(scale=1, rotate=1, aniso=1, xfrac=1), p=0.8, shape=[4, 3, 36, 28], seed=15),
    dict(name='color',         cfg=dict(brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1), p=0.9, shape=[5, 3, 16, 20], seed=16),
    dict(name='color_gray',    cfg=dict(brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1), p=1.0, shape=[4, 1, 16, 16], seed=17),
    dict(name='bgcfnc',        cfg=dict(BGC, imgfilter=1, noise=1, cutout=1), p=0.7, shape=[4, 3, 48, 48], seed=18),
    dict(name='filter_bands',  cfg=dict(imgfilter=1, imgfilter_bands=[1, 0, 0.5, 1]), p=1.0, shape=[3, 3, 32, 40], seed=19),
    dict(name='pct30_bgc',     cfg=BGC, p=1.0, shape=[2, 3, 40, 40], seed=20, debug_percentile=0.3),
    dict(name='pct80_bgcfnc',  cfg=dict(BGC, imgfilter=1, noise=1, cutout=1), p=1.0, shape=[2, 3, 40, 48], seed=21, debug_percentile=0.8),
    dict(name='rotate_only',   cfg=dict(rotate=1, rotate_max=0.25), p=1.0, shape=[4, 3, 32, 32], seed=22),
]


class Recorder:
    """Wraps torch.rand / torch.randn: same numbers as the originals, every call kept."""
    def __init__(self):
        self.calls = []
        self._rand, self._randn = torch.rand, torch.randn
    def __enter__(self):
        def rand(*a, **k):
            t = self._rand(*a, **k); self.calls.append(('u', t.clone())); return t
        def randn(*a, **k):
            t = self._randn(*a, **k); self.calls.append(('z', t.clone())); return t
        torch.rand, torch.randn = rand, randn
        return self
    def __exit__(self, *exc):
        torch.rand, torch.randn = self._rand, self._randn


def lay_out(calls, cfg, n, channels):
    """Recorded calls (in the reference's order) -> u [n,29], z [n,12], noise field."""
    cfg = {**RA.DEFAULTS, **cfg}
    u = torch.full([n, len(RA.U_COLS)], 0.5)
    z = torch.zeros([n, len(RA.Z_COLS)])
    field = None
    calls = list(calls)
    for transform, draws in RA.DRAW_ORDER:
        if not cfg[transform] > 0 or (transform in ('hue', 'saturation') and channels == 1):
            continue
        for kind, cols in draws:
            got_kind, t = calls.pop(0)
            if kind == 'field':
                assert got_kind == 'z' and t.ndim == 4
                field = t
                continue
            assert got_kind == kind, (transform, cols, got_kind)
            cols = cols if isinstance(cols, tuple) else (cols,)
            t = t.reshape(n, len(cols))
            table, names = (u, RA.U_COLS) if kind == 'u' else (z, RA.Z_COLS)
            for j, c in enumerate(cols):
                table[:, names.index(c)] = t[:, j]
    assert not calls, f'{len(calls)} recorded draws left over'
    return u, z, field


def gen_augment(ref_root):
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    os.chdir(ref_root)
    import training.augment as ref_aug
    out = {}
    pipe = ref_aug.AugmentPipe()
    out['Hz_fbank'] = pipe.Hz_fbank.numpy()
    out['Hz_geom'] = pipe.Hz_geom.numpy()
    for case in CASES:
        n, ch, h, w = case['shape']
        g = torch.Generator().manual_seed(case['seed'])
        # smooth-ish images so that the resampling has structure to move: low-frequency field + noise
        base = torch.nn.functional.interpolate(torch.randn([n, ch, 6, 6], generator=g), size=[h, w], mode='bicubic', align_corners=False)
        images = (base + 0.3 * torch.randn([n, ch, h, w], generator=g)).clamp(-1.5, 1.5).requires_grad_(True)
        probe = torch.randn([n, ch, h, w], generator=g)
        pipe = ref_aug.AugmentPipe(**case['cfg'])
        pipe.p.copy_(torch.as_tensor(case['p']))
        torch.manual_seed(case['seed'] * 7 + 1)
        with Recorder() as rec:
            y = pipe(images, debug_percentile=case.get('debug_percentile'))
        dx, = torch.autograd.grad((y * probe).sum(), images)
        u, z, field = lay_out(rec.calls, case['cfg'], n, ch)
        k = case['name']
        out[k + '.x'], out[k + '.probe'] = images.detach().numpy(), probe.numpy()
        out[k + '.u'], out[k + '.z'] = u.numpy(), z.numpy()
        if field is not None:
            out[k + '.field'] = field.numpy()
        out[k + '.y'], out[k + '.dx'] = y.detach().numpy(), dx.numpy()
        # the restatement against the reference, on the spot
        mine = RA.augment(images.detach(), u, z, case['cfg'], case['p'], noise_field=field, debug_percentile=case.get('debug_percentile'))
        err = float((mine - y.detach()).abs().max() / y.detach().abs().max())
        print(f"{k:16s} y {tuple(y.shape)} |y| {float(y.abs().max()):.3f}  restatement max-rel-err {err:.2e}", flush=True)
    np.savez_compressed(os.path.join(GOLDEN, 'augment.npz'), **out)
    print('augment fixtures written:', len(out), 'arrays')
