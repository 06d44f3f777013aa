"""Deterministic weights and inputs shared by the golden-vector generator and the tests -- TEST
INFRASTRUCTURE. Values are a closed form of (tensor name, element index), so the reference modules
(in the build container) and this repository's modules (anywhere) get identical parameters without
shipping a state dict or depending on RNG call order."""

import zlib

import numpy as np
import torch


def _wave(name, numel, scale):
    idx = np.arange(numel, dtype=np.float64)
    h = zlib.crc32(name.encode()) % 10007
    v = np.sin(idx * 0.618033988749895 + h * 0.37) + 0.5 * np.sin(idx * 0.0137 + h)
    return (v * scale).astype(np.float32)


def _normal(name, numel, scale):
    """Standard-normal values that depend only on (tensor name, element index): numpy's legacy MT19937 stream seeded by the
    CRC of the name (stable across numpy versions and platforms by numpy's compatibility guarantee for RandomState)."""
    return (np.random.RandomState(zlib.crc32(name.encode())).standard_normal(numel) * scale).astype(np.float32)


def fill_module(module, kind='wave'):
    """Overwrite every parameter and buffer of ``module`` in place (resample filters are left alone).
    ``kind='wave'``: smooth closed form (sums over wide layers largely cancel: a full-width discriminator filled this way
    barely depends on its image).  ``kind='normal'``: unit-variance pseudo-random weights, the statistics of the
    reference's own initialisation (``torch.randn`` weights, networks.py:149), so signals and gradients keep O(1) size
    through 512-channel layers -- used where a derivative with respect to the IMAGE must be non-degenerate (R1)."""
    gen = _wave if kind == 'wave' else _normal
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if name.endswith('resample_filter') or name.endswith('w_avg') or t.numel() == 0:
                continue
            if name.endswith('noise_strength'):
                t.fill_(0.1)
                continue
            scale = 0.9 if kind == 'wave' else 1.0
            if kind == 'normal' and '.'.join(name.split('.')[-3:-1]).startswith('mapping.fc') and name.endswith('weight'):
                scale = 100.0       # mapping layers run at lr_multiplier 0.01: the reference initialises them as randn / 0.01 (networks.py:108)
            if name.endswith('bias') or name.endswith('bias1'):
                scale = 0.1
            t.copy_(torch.from_numpy(gen(name, t.numel(), scale)).reshape(t.shape))
    return module


def make_inputs(n=2, seed=0, res=256):
    """Synthetic batch with the shapes of training_loop_wo_flow_fullbody.py:425-456."""
    g = torch.Generator().manual_seed(1234 + seed)
    def u(*shape):
        return torch.rand(shape, generator=g) * 2 - 1
    def blobs(p):
        coarse = torch.rand([n, 1, res // 16, res // 16], generator=g)
        return (torch.nn.functional.interpolate(coarse, size=(res, res), mode='bilinear', align_corners=False) < p).float()
    real_img = u(n, 3, res, res)
    real_img[..., : res // 8] = 1.0
    real_img[..., res - res // 8:] = 1.0
    mask = blobs(0.5)
    retain = mask * real_img - (1 - mask)
    lines = (torch.rand([n, 3, res, res], generator=g) < 0.02).float() * 2 - 1
    pose = torch.cat([lines, retain], dim=1)
    style = u(n, 42, res // 4, res // 4)
    drop = (torch.rand([n, 14, 1, 1], generator=g) < 0.3).repeat_interleave(3, dim=1)
    style = torch.where(drop, -torch.ones_like(style), style)
    du_mask, dl_mask = blobs(0.35), blobs(0.35)
    du_in = u(n, 3, res, res) * du_mask - (1 - du_mask)
    dl_in = u(n, 3, res, res) * dl_mask - (1 - dl_mask)
    gt_parsing = torch.randint(0, 6, [n, 1, res // 8, res // 8], generator=g).float()
    gt_parsing = torch.nn.functional.interpolate(gt_parsing, size=(res, res), mode='nearest')
    return dict(real_img=real_img, pose=pose, style_input=style, retain=retain, denorm_upper_input=du_in,
                denorm_lower_input=dl_in, denorm_upper_mask=du_mask, denorm_lower_mask=dl_mask, gt_parsing=gt_parsing,
                gen_z=torch.zeros([n, 0]))


# Configuration of the golden model runs: the 'fashion' preset narrowed to channel_base 2048
# (encoders and SPADE feature widths are fixed by the architecture and stay full width).
G_KWARGS = dict(z_dim=0, c_dim=512, w_dim=512, img_resolution=256, img_channels=3, mapping_kwargs=dict(num_layers=1),
                synthesis_kwargs=dict(channel_base=2048, channel_max=512, conv_clamp=256))
D_KWARGS = dict(c_dim=512, img_resolution=256, img_channels=3, channel_base=2048, channel_max=512, conv_clamp=256)


def summarize(t, samples=4096):
    """Compact fingerprint of a tensor: strided samples + moments (what the golden files store)."""
    t = t.detach().float().cpu()
    flat = t.reshape(-1)
    step = max(flat.numel() // samples, 1)
    return dict(sample=flat[::step][:samples].numpy().copy(),
                moments=np.array([flat.sum().item(), flat.abs().sum().item(), flat.square().sum().item()], dtype=np.float64))
