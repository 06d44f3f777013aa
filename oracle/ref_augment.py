"""CPU restatement of the reference's ADA augmentation pipeline -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` (and ``oracle/make_golden_augment.py``) import this module; nothing under ``pasta-gan_amd/`` does.

``augment(images, u, z, cfg, p, ...)`` restates ``AugmentPipe.forward`` (training/augment.py:174-431 of the reference) as a
function of EXPLICIT random numbers: ``u [N, 29]`` uniform and ``z [N, 12]`` normal, columns ``U_COLS`` / ``Z_COLS`` in the
order the reference draws them, plus the additive-noise field.  Parity status: PINNED -- ``oracle/make_golden_augment.py``
runs the reference's own ``AugmentPipe`` on the CPU with its ``torch.rand`` / ``torch.randn`` calls recorded, stores the
recorded numbers and the reference's outputs in ``tests/golden/augment.npz``; ``tests/test_augment.py`` replays them
through this file.  Stock PyTorch CPU ops in float32 (as the reference), autograd-able in the images.
"""

import numpy as np
import torch

from oracle import ref_ops as RO

SYM2 = [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025]
SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
        0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
        0.0017677118642428036, -0.007800708325034148]

# (transform, what) in drawing order; 'on' = the uniform compared against multiplier * p
U_COLS = ['xflip.i', 'xflip.on', 'rotate90.i', 'rotate90.on', 'xint.x', 'xint.y', 'xint.on', 'scale.on',
          'rotate.pre', 'rotate.pre.on', 'aniso.on', 'rotate.post', 'rotate.post.on', 'xfrac.on',
          'brightness.on', 'contrast.on', 'lumaflip.i', 'lumaflip.on', 'hue', 'hue.on', 'saturation.on',
          'imgfilter.on.0', 'imgfilter.on.1', 'imgfilter.on.2', 'imgfilter.on.3', 'noise.on', 'cutout.on',
          'cutout.x', 'cutout.y']
Z_COLS = ['scale', 'aniso', 'xfrac.x', 'xfrac.y', 'brightness', 'contrast', 'saturation',
          'imgfilter.0', 'imgfilter.1', 'imgfilter.2', 'imgfilter.3', 'noise.sigma']

DEFAULTS = dict(xflip=0, rotate90=0, xint=0, xint_max=0.125, scale=0, rotate=0, aniso=0, xfrac=0, scale_std=0.2, rotate_max=1,
                aniso_std=0.2, xfrac_std=0.125, brightness=0, contrast=0, lumaflip=0, hue=0, saturation=0, brightness_std=0.2,
                contrast_std=0.5, hue_max=1, saturation_std=1, imgfilter=0, imgfilter_bands=[1, 1, 1, 1], imgfilter_std=1,
                noise=0, cutout=0, noise_std=0.1, cutout_size=0.5)

# The order in which an enabled transform consumes the generator (kind, shape-per-batch); used by the golden script to
# lay the reference's recorded draws into the columns above.  augment.py:192-261, 311-348, 375-405.
DRAW_ORDER = [
    ('xflip',      [('u', 'xflip.i'), ('u', 'xflip.on')]),
    ('rotate90',   [('u', 'rotate90.i'), ('u', 'rotate90.on')]),
    ('xint',       [('u', ('xint.x', 'xint.y')), ('u', 'xint.on')]),
    ('scale',      [('z', 'scale'), ('u', 'scale.on')]),
    ('rotate',     [('u', 'rotate.pre'), ('u', 'rotate.pre.on')]),
    ('aniso',      [('z', 'aniso'), ('u', 'aniso.on')]),
    ('rotate',     [('u', 'rotate.post'), ('u', 'rotate.post.on')]),
    ('xfrac',      [('z', ('xfrac.x', 'xfrac.y')), ('u', 'xfrac.on')]),
    ('brightness', [('z', 'brightness'), ('u', 'brightness.on')]),
    ('contrast',   [('z', 'contrast'), ('u', 'contrast.on')]),
    ('lumaflip',   [('u', 'lumaflip.i'), ('u', 'lumaflip.on')]),
    ('hue',        [('u', 'hue'), ('u', 'hue.on')]),
    ('saturation', [('z', 'saturation'), ('u', 'saturation.on')]),
    ('imgfilter',  [('z', 'imgfilter.0'), ('u', 'imgfilter.on.0'), ('z', 'imgfilter.1'), ('u', 'imgfilter.on.1'),
                    ('z', 'imgfilter.2'), ('u', 'imgfilter.on.2'), ('z', 'imgfilter.3'), ('u', 'imgfilter.on.3')]),
    ('noise',      [('z', 'noise.sigma'), ('u', 'noise.on'), ('field', None)]),
    ('cutout',     [('u', 'cutout.on'), ('u', ('cutout.x', 'cutout.y'))]),
]


def filter_bank():
    """[4, 43] band-pass bank (augment.py:162-172): band 0 = everything below the three octave bands 1..3."""
    lo = np.asarray(SYM2)
    hi = lo * np.where(np.arange(lo.size) % 2 == 0, 1.0, -1.0)
    lo2, hi2 = np.convolve(lo, lo[::-1]) / 2, np.convolve(hi, hi[::-1]) / 2
    bank = np.zeros([4, 1]); bank[0, 0] = 1
    for band in (1, 2, 3):
        wide = np.zeros([4, 2 * bank.shape[1] - 1])
        wide[:, 0::2] = bank
        bank = np.array([np.convolve(r, lo2) for r in wide])
        c = (bank.shape[1] - hi2.size) // 2
        bank[band, c:c + hi2.size] += hi2
    return bank


def _affine(n, rows):
    """Batched homogeneous matrix [n, k, k] from rows of scalars / [n] tensors."""
    out = []
    for row in rows:
        out.append(torch.stack([e.to(torch.float32) if isinstance(e, torch.Tensor) else torch.full([n], float(e)) for e in row], dim=-1))
    return torch.stack(out, dim=-2)

def _rot(n, th):
    return _affine(n, [[torch.cos(th), torch.sin(-th), 0], [torch.sin(th), torch.cos(th), 0], [0, 0, 1]])
def _scl(n, sx, sy):
    return _affine(n, [[sx, 0, 0], [0, sy, 0], [0, 0, 1]])
def _shf(n, tx, ty):
    return _affine(n, [[1, 0, tx], [0, 1, ty], [0, 0, 1]])


def geometry_matrix(u, z, cfg, p, width, height, dp=None):
    """G_inv [N,3,3] (pixel_out -> pixel_in), augment.py:186-263; None when no geometric transform is enabled."""
    n = u.shape[0]
    U = lambda name: u[:, U_COLS.index(name)]
    Z = lambda name: z[:, Z_COLS.index(name)]
    on = lambda name, prob: U(name) < prob
    pct = None if dp is None else torch.as_tensor(dp, dtype=torch.float32)
    G = None
    def then(M):
        return M if G is None else G @ M
    if cfg['xflip'] > 0:
        i = torch.where(on('xflip.on', cfg['xflip'] * p), torch.floor(U('xflip.i') * 2), torch.zeros(n))
        if pct is not None: i = torch.full([n], float(torch.floor(pct * 2)))
        G = then(_scl(n, 1 / (1 - 2 * i), 1))
    if cfg['rotate90'] > 0:
        i = torch.where(on('rotate90.on', cfg['rotate90'] * p), torch.floor(U('rotate90.i') * 4), torch.zeros(n))
        if pct is not None: i = torch.full([n], float(torch.floor(pct * 4)))
        G = then(_rot(n, -(-np.pi / 2 * i)))
    if cfg['xint'] > 0:
        t = torch.stack([U('xint.x'), U('xint.y')], dim=1) * 2 - 1
        t = torch.where(on('xint.on', cfg['xint'] * p)[:, None], t * cfg['xint_max'], torch.zeros(n, 2))
        if pct is not None: t = torch.full([n, 2], float((pct * 2 - 1) * cfg['xint_max']))
        G = then(_shf(n, -torch.round(t[:, 0] * width), -torch.round(t[:, 1] * height)))
    if cfg['scale'] > 0:
        s = torch.where(on('scale.on', cfg['scale'] * p), torch.exp2(Z('scale') * cfg['scale_std']), torch.ones(n))
        if pct is not None: s = torch.full([n], float(torch.exp2(torch.erfinv(pct * 2 - 1) * cfg['scale_std'])))
        G = then(_scl(n, 1 / s, 1 / s))
    p_rot = 1 - torch.sqrt((1 - cfg['rotate'] * p).clamp(0, 1))
    if cfg['rotate'] > 0:
        th = torch.where(on('rotate.pre.on', p_rot), (U('rotate.pre') * 2 - 1) * np.pi * cfg['rotate_max'], torch.zeros(n))
        if pct is not None: th = torch.full([n], float((pct * 2 - 1) * np.pi * cfg['rotate_max']))
        G = then(_rot(n, th))
    if cfg['aniso'] > 0:
        s = torch.where(on('aniso.on', cfg['aniso'] * p), torch.exp2(Z('aniso') * cfg['aniso_std']), torch.ones(n))
        if pct is not None: s = torch.full([n], float(torch.exp2(torch.erfinv(pct * 2 - 1) * cfg['aniso_std'])))
        G = then(_scl(n, 1 / s, s))
    if cfg['rotate'] > 0:
        th = torch.where(on('rotate.post.on', p_rot), (U('rotate.post') * 2 - 1) * np.pi * cfg['rotate_max'], torch.zeros(n))
        if pct is not None: th = torch.zeros(n)
        G = then(_rot(n, th))
    if cfg['xfrac'] > 0:
        t = torch.stack([Z('xfrac.x'), Z('xfrac.y')], dim=1) * cfg['xfrac_std']
        t = torch.where(on('xfrac.on', cfg['xfrac'] * p)[:, None], t, torch.zeros(n, 2))
        if pct is not None: t = torch.full([n, 2], float(torch.erfinv(pct * 2 - 1) * cfg['xfrac_std']))
        G = then(_shf(n, -t[:, 0] * width, -t[:, 1] * height))
    return G


def color_matrix(u, z, cfg, p, channels, dp=None):
    """C [N,4,4] (colour_in -> colour_out), augment.py:306-350; None when no colour transform is enabled."""
    n = u.shape[0]
    U = lambda name: u[:, U_COLS.index(name)]
    Z = lambda name: z[:, Z_COLS.index(name)]
    on = lambda name, prob: U(name) < prob
    pct = None if dp is None else torch.as_tensor(dp, dtype=torch.float32)
    luma = torch.tensor([1, 1, 1, 0], dtype=torch.float32) / np.sqrt(3)
    P = torch.outer(luma, luma)                                  # projector on the luma axis
    I = torch.eye(4)
    C = None
    def after(M):
        return M if C is None else M @ C
    if cfg['brightness'] > 0:
        b = torch.where(on('brightness.on', cfg['brightness'] * p), Z('brightness') * cfg['brightness_std'], torch.zeros(n))
        if pct is not None: b = torch.full([n], float(torch.erfinv(pct * 2 - 1) * cfg['brightness_std']))
        C = after(_affine(n, [[1, 0, 0, b], [0, 1, 0, b], [0, 0, 1, b], [0, 0, 0, 1]]))
    if cfg['contrast'] > 0:
        c = torch.where(on('contrast.on', cfg['contrast'] * p), torch.exp2(Z('contrast') * cfg['contrast_std']), torch.ones(n))
        if pct is not None: c = torch.full([n], float(torch.exp2(torch.erfinv(pct * 2 - 1) * cfg['contrast_std'])))
        C = after(_affine(n, [[c, 0, 0, 0], [0, c, 0, 0], [0, 0, c, 0], [0, 0, 0, 1]]))
    if cfg['lumaflip'] > 0:
        i = torch.where(on('lumaflip.on', cfg['lumaflip'] * p), torch.floor(U('lumaflip.i') * 2), torch.zeros(n))
        if pct is not None: i = torch.full([n], float(torch.floor(pct * 2)))
        C = after(I - 2 * P * i[:, None, None])
    if cfg['hue'] > 0 and channels > 1:
        th = torch.where(on('hue.on', cfg['hue'] * p), (U('hue') * 2 - 1) * np.pi * cfg['hue_max'], torch.zeros(n))
        if pct is not None: th = torch.full([n], float((pct * 2 - 1) * np.pi * cfg['hue_max']))
        # Rodrigues: R = cos I + sin [v]x + (1 - cos) v v^T around the luma axis
        v = luma[:3]
        a = float(v[0])
        K = torch.tensor([[0, -a, a], [a, 0, -a], [-a, a, 0]], dtype=torch.float32)
        R3 = torch.cos(th)[:, None, None] * torch.eye(3) + torch.sin(th)[:, None, None] * K + (1 - torch.cos(th))[:, None, None] * torch.outer(v, v)
        R = torch.eye(4).repeat(n, 1, 1)
        R[:, :3, :3] = R3
        C = after(R)
    if cfg['saturation'] > 0 and channels > 1:
        s = torch.where(on('saturation.on', cfg['saturation'] * p), torch.exp2(Z('saturation') * cfg['saturation_std']), torch.ones(n))
        if pct is not None: s = torch.full([n], float(torch.exp2(torch.erfinv(pct * 2 - 1) * cfg['saturation_std'])))
        C = after(P + (I - P) * s[:, None, None])
    return C


def augment(images, u, z, cfg, p, noise_field=None, debug_percentile=None):
    cfg = {**DEFAULTS, **cfg}
    n, ch, h, w = images.shape
    p = torch.as_tensor(p, dtype=torch.float32)
    f_geom = RO.setup_filter(SYM6)
    hz_pad = f_geom.shape[0] // 4

    G = geometry_matrix(u, z, cfg, p, w, h, debug_percentile)
    if G is not None:       # augment.py:268-301
        cx, cy = (w - 1) / 2, (h - 1) / 2
        corners = torch.tensor([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], dtype=torch.float32)
        moved = G @ corners.t()                                                 # [n, xyz, corner]
        xs, ys = moved[:, 0, :].flatten(), moved[:, 1, :].flatten()
        reach = torch.stack([(-xs).max(), (-ys).max(), xs.max(), ys.max()])
        reach = reach + torch.tensor([hz_pad * 2 - cx, hz_pad * 2 - cy] * 2, dtype=torch.float32)
        reach = torch.minimum(reach.clamp(min=0), torch.tensor([w - 1, h - 1] * 2, dtype=torch.float32))
        mx0, my0, mx1, my1 = [int(v) for v in reach.ceil()]
        images = torch.nn.functional.pad(images, [mx0, mx1, my0, my1], mode='reflect')
        G = _shf(1, (mx0 - mx1) / 2, (my0 - my1) / 2) @ G
        images = RO.upsample2d(images, f_geom, up=2)
        G = _scl(1, 2, 2) @ G @ _scl(1, 0.5, 0.5)
        G = _shf(1, -0.5, -0.5) @ G @ _shf(1, 0.5, 0.5)
        out_h, out_w = (h + hz_pad * 2) * 2, (w + hz_pad * 2) * 2
        G = _scl(1, 2 / images.shape[3], 2 / images.shape[2]) @ G @ _scl(1, out_w / 2, out_h / 2)
        grid = torch.nn.functional.affine_grid(G[:, :2, :], [n, ch, out_h, out_w], align_corners=False)
        images = torch.nn.functional.grid_sample(images, grid, mode='bilinear', padding_mode='zeros', align_corners=False)
        images = RO.downsample2d(images, f_geom, down=2, padding=-hz_pad * 2, flip_filter=True)

    C = color_matrix(u, z, cfg, p, ch, debug_percentile)
    if C is not None:       # augment.py:354-364
        flat = images.reshape(n, ch, h * w)
        if ch == 3:
            flat = C[:, :3, :3] @ flat + C[:, :3, 3:]
        elif ch == 1:
            row = C[:, :3, :].mean(dim=1, keepdim=True)
            flat = flat * row[:, :, :3].sum(dim=2, keepdim=True) + row[:, :, 3:]
        else:
            raise ValueError('Image must be RGB (3 channels) or L (1 channel)')
        images = flat.reshape(n, ch, h, w)

    if cfg['imgfilter'] > 0:    # augment.py:370-400
        bank = torch.as_tensor(filter_bank(), dtype=torch.float32)
        power = torch.tensor([10, 1, 1, 1], dtype=torch.float32) / 13
        gains = torch.ones(n, 4)
        for band, strength in enumerate(cfg['imgfilter_bands']):
            a = torch.exp2(z[:, Z_COLS.index(f'imgfilter.{band}')] * cfg['imgfilter_std'])
            a = torch.where(u[:, U_COLS.index(f'imgfilter.on.{band}')] < cfg['imgfilter'] * p * strength, a, torch.ones(n))
            if debug_percentile is not None:
                pct = torch.as_tensor(debug_percentile, dtype=torch.float32)
                a = torch.full([n], float(torch.exp2(torch.erfinv(pct * 2 - 1) * cfg['imgfilter_std']))) if strength > 0 else torch.ones(n)
            t = torch.ones(n, 4)
            t[:, band] = a
            gains = gains * (t / (power * t * t).sum(dim=1, keepdim=True).sqrt())
        taps = gains @ bank                                                     # [n, 43]
        half = bank.shape[1] // 2
        x = torch.nn.functional.pad(images.reshape(1, n * ch, h, w), [half] * 4, mode='reflect')
        k = taps.repeat_interleave(ch, dim=0)                                   # one filter per (sample, channel) plane
        x = torch.nn.functional.conv2d(x, k[:, None, None, :], groups=n * ch)
        x = torch.nn.functional.conv2d(x, k[:, None, :, None], groups=n * ch)
        images = x.reshape(n, ch, h, w)

    if cfg['noise'] > 0:        # augment.py:406-412
        sigma = z[:, Z_COLS.index('noise.sigma')].abs() * cfg['noise_std']
        sigma = torch.where(u[:, U_COLS.index('noise.on')] < cfg['noise'] * p, sigma, torch.zeros(n))
        if debug_percentile is not None:
            sigma = torch.full([n], float(torch.erfinv(torch.as_tensor(debug_percentile, dtype=torch.float32)) * cfg['noise_std']))
        images = images + noise_field * sigma[:, None, None, None]

    if cfg['cutout'] > 0:       # augment.py:414-428
        size = torch.where(u[:, U_COLS.index('cutout.on')] < cfg['cutout'] * p, torch.full([n], cfg['cutout_size']), torch.zeros(n))
        cxs, cys = u[:, U_COLS.index('cutout.x')], u[:, U_COLS.index('cutout.y')]
        if debug_percentile is not None:
            size = torch.full([n], cfg['cutout_size'])
            cxs = cys = torch.full([n], float(debug_percentile))
        col = (torch.arange(w) + 0.5) / w
        row = (torch.arange(h) + 0.5) / h
        outside_x = (col[None, :] - cxs[:, None]).abs() >= size[:, None] / 2                     # [n, w]
        outside_y = (row[None, :] - cys[:, None]).abs() >= size[:, None] / 2                     # [n, h]
        keep = outside_y[:, :, None] | outside_x[:, None, :]
        images = images * keep[:, None].to(torch.float32)

    return images
