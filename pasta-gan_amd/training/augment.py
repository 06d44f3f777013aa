"""ADA augmentation pipeline for the discriminator's inputs (reference: training/augment.py).

Same class name, constructor arguments, buffers (``p``, ``Hz_geom``, ``Hz_fbank``) and ``forward(images,
debug_percentile=None)`` as the reference's ``AugmentPipe`` (augment.py:121-172), same distributions and the same
image operations, arranged for the GPU instead of a long chain of tiny tensor ops:

* every random parameter of a call comes from two draws, ``u = rand([N, 29])`` and ``z = randn([N, 12])``, whose columns
  are listed in ``DRAWS_U`` / ``DRAWS_Z`` in the order the reference draws them (so a test can replay the reference's
  stream); one HIP launch (``pasta_ada_matrices``) turns them into the per-sample inverse geometric transform, the
  per-sample colour matrix and the reflect-padding margins of the batch (the reference: ~150 launches);
* the margins size the padded tensor, so they are read back (one 16-byte copy; the reference synchronises at the same
  point, augment.py:282); the pad / 2x upsampling / sampling-grid bookkeeping matrices are then host constants and fold
  into one more tiny launch (``pasta_ada_theta``);
* 2x upsampling and the final 2x decimation run on the HIP ``upfirdn2d`` (separable 12-tap ``sym6``), the bilinear
  resampling on ``grid_sample_gradfix.affine_sample`` (``pasta_affine_sample``: no grid tensor; its gradient is a gather,
  free of atomics), the colour transform on ``pasta_color_affine`` (one pass over the images).

There is no CPU path: the CPU restatement the parity tests use is ``oracle/ref_augment.py``.
"""

import ctypes

import numpy as np
import torch

from torch_utils import persistence
from torch_utils import custom_ops
from torch_utils.ops import _native
from torch_utils.ops import upfirdn2d
from torch_utils.ops import grid_sample_gradfix
from torch_utils.ops import conv2d_gradfix

#----------------------------------------------------------------------------
# Low-pass decomposition filters (Daubechies least-asymmetric wavelets; augment.py:21-39 holds the whole family,
# these are the two the pipeline uses).

wavelets = {
    'sym2': [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025],
    'sym6': [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
             0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
             0.0017677118642428036, -0.007800708325034148],
}

# Columns of the uniform and of the normal draw, in the order augment.py draws them (:192-261, :311-348, :375-405).
DRAWS_U = ['xflip.i', 'xflip.on', 'rotate90.i', 'rotate90.on', 'xint.x', 'xint.y', 'xint.on', 'scale.on',
           'rotate.pre', 'rotate.pre.on', 'aniso.on', 'rotate.post', 'rotate.post.on', 'xfrac.on',
           'brightness.on', 'contrast.on', 'lumaflip.i', 'lumaflip.on', 'hue', 'hue.on', 'saturation.on',
           'imgfilter.on.0', 'imgfilter.on.1', 'imgfilter.on.2', 'imgfilter.on.3', 'noise.on', 'cutout.on',
           'cutout.x', 'cutout.y']
DRAWS_Z = ['scale', 'aniso', 'xfrac.x', 'xfrac.y', 'brightness', 'contrast', 'saturation',
           'imgfilter.0', 'imgfilter.1', 'imgfilter.2', 'imgfilter.3', 'noise.sigma']

_KERNEL_FIELDS = [name for name, _ in custom_ops.AdaConfig._fields_]

#----------------------------------------------------------------------------

def _filter_bank():
    """Band-pass filters of the image-space filtering step: rows = 4 octave bands, 2-tap-per-octave dilations of the
    ``sym2`` half-band pair (augment.py:162-172)."""
    lo = np.asarray(wavelets['sym2'])
    hi = lo * ((-1) ** np.arange(lo.size))
    lo2 = np.convolve(lo, lo[::-1]) / 2
    hi2 = np.convolve(hi, hi[::-1]) / 2
    bank = np.eye(4, 1)
    for i in range(1, bank.shape[0]):
        dilated = np.zeros([bank.shape[0], bank.shape[1] * 2 - 1])
        dilated[:, ::2] = bank                                      # insert a zero between taps
        bank = np.stack([np.convolve(row, lo2) for row in dilated])
        mid = bank.shape[1] // 2
        bank[i, mid - hi2.size // 2: mid - hi2.size // 2 + hi2.size] += hi2
    return bank

def _mat3(kind, a, b):
    m = np.eye(3)
    if kind == 'scale':
        m[0, 0], m[1, 1] = a, b
    else:
        m[0, 2], m[1, 2] = a, b
    return m

class _ColorAffine(torch.autograd.Function):
    """images [N,3,H,W], C [N,4,4] -> C[:, :3, :3] @ images + C[:, :3, 3:] (mode 0).  Linear in the images: the gradient
    is the kernel's adjoint mode (1), the gradient of that its linear-part mode (2), and so on alternately."""
    @staticmethod
    def forward(ctx, images, C, mode):
        images = images.contiguous()
        out = torch.empty_like(images)
        n, _, h, w = images.shape
        with torch.cuda.device(images.device):
            st = _native.lib().pasta_color_affine(_native.ptr(images), _native.ptr(C), _native.ptr(out), n, h * w, mode, _native.stream())
        _native.check(st)
        ctx.save_for_backward(C)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, dout):
        C, = ctx.saved_tensors
        dx = _ColorAffine.apply(dout, C, 2 if ctx.mode == 1 else 1) if ctx.needs_input_grad[0] else None
        return dx, None, None

#----------------------------------------------------------------------------

@persistence.persistent_class
class AugmentPipe(torch.nn.Module):
    """augment.py:121-431.  All augmentations are off by default; a multiplier of 1 enables one."""
    def __init__(self,
        xflip=0, rotate90=0, xint=0, xint_max=0.125,
        scale=0, rotate=0, aniso=0, xfrac=0, scale_std=0.2, rotate_max=1, aniso_std=0.2, xfrac_std=0.125,
        brightness=0, contrast=0, lumaflip=0, hue=0, saturation=0, brightness_std=0.2, contrast_std=0.5, hue_max=1, saturation_std=1,
        imgfilter=0, imgfilter_bands=[1,1,1,1], imgfilter_std=1,
        noise=0, cutout=0, noise_std=0.1, cutout_size=0.5,
    ):
        super().__init__()
        self.register_buffer('p', torch.ones([]))       # overall multiplier of every probability below
        given = dict(locals())
        for name in ['xflip', 'rotate90', 'xint', 'xint_max', 'scale', 'rotate', 'aniso', 'xfrac', 'scale_std', 'rotate_max',
                     'aniso_std', 'xfrac_std', 'brightness', 'contrast', 'lumaflip', 'hue', 'saturation', 'brightness_std',
                     'contrast_std', 'hue_max', 'saturation_std', 'imgfilter', 'imgfilter_std', 'noise', 'cutout', 'noise_std',
                     'cutout_size']:
            setattr(self, name, float(given[name]))
        self.imgfilter_bands = list(imgfilter_bands)
        self.register_buffer('Hz_geom', upfirdn2d.setup_filter(wavelets['sym6']))           # orthogonal low-pass of the geometric step
        self.register_buffer('Hz_fbank', torch.as_tensor(_filter_bank(), dtype=torch.float32))

    # -- parameters -------------------------------------------------------------------------------------------------

    def _kernel_config(self):
        return custom_ops.AdaConfig(*[getattr(self, name) for name in _KERNEL_FIELDS])

    def _has_geometry(self):
        return any(getattr(self, k) > 0 for k in ['xflip', 'rotate90', 'xint', 'scale', 'rotate', 'aniso', 'xfrac'])

    def _has_color(self, num_channels):
        names = ['brightness', 'contrast', 'lumaflip'] + (['hue', 'saturation'] if num_channels > 1 else [])
        return any(getattr(self, k) > 0 for k in names)

    def draw(self, batch_size, device):
        """The random numbers of one call: dict(u=[N, 29] uniform, z=[N, 12] normal)."""
        return dict(u=torch.rand([batch_size, len(DRAWS_U)], device=device), z=torch.randn([batch_size, len(DRAWS_Z)], device=device))

    def matrices(self, draws, width, height, num_channels, debug_percentile=None):
        """-> G_inv [N,3,3], C [N,4,4], margins int32 [4] (device tensors)."""
        u, z = draws['u'].contiguous(), draws['z'].contiguous()
        n, dev = u.shape[0], u.device
        G_inv = torch.empty([n, 3, 3], device=dev)
        C = torch.empty([n, 4, 4], device=dev)
        margins = torch.empty([4], dtype=torch.int32, device=dev)
        cfg = self._kernel_config()
        dp = -1.0 if debug_percentile is None else float(debug_percentile)
        with torch.cuda.device(dev):
            st = _native.lib().pasta_ada_matrices(_native.ptr(u), _native.ptr(z), n, u.shape[1], z.shape[1], _native.ptr(self.p),
                                                  ctypes.byref(cfg), width, height, num_channels, self.Hz_geom.shape[0] // 4, dp,
                                                  _native.ptr(G_inv), _native.ptr(C), _native.ptr(margins), _native.stream())
        _native.check(st)
        return G_inv, C, margins

    # -- the pipeline -----------------------------------------------------------------------------------------------

    def forward(self, images, debug_percentile=None, draws=None):
        assert isinstance(images, torch.Tensor) and images.ndim == 4
        _native.require_gpu(images, 'AugmentPipe')
        if images.dtype != torch.float32:
            raise RuntimeError('AugmentPipe: float32 images only')
        batch_size, num_channels, height, width = images.shape
        device = images.device
        if debug_percentile is not None:
            debug_percentile = float(debug_percentile)
        if draws is None:
            draws = self.draw(batch_size, device)
        geometry, color = self._has_geometry(), self._has_color(num_channels)
        if geometry or color:
            G_inv, C, margins = self.matrices(draws, width, height, num_channels, debug_percentile)

        if geometry:        # pad (reflect) -> 2x up -> resample under G_inv -> 2x down and crop (augment.py:268-301)
            mx0, my0, mx1, my1 = margins.tolist()
            hz_pad = self.Hz_geom.shape[0] // 4
            images = torch.nn.functional.pad(input=images, pad=[mx0, mx1, my0, my1], mode='reflect')
            images = upfirdn2d.upsample2d(x=images, f=self.Hz_geom, up=2)
            out_h, out_w = (height + hz_pad * 2) * 2, (width + hz_pad * 2) * 2
            # pixel_out -> pixel_in in normalised coordinates of the sampling grid: A @ G_inv @ B with
            #   A = to-normalised(in) . half-pixel shift . 2x . origin shift of the padding
            #   B = 1/2x . half-pixel shift back . from-normalised(out)
            A = _mat3('scale', 2 / images.shape[3], 2 / images.shape[2]) @ _mat3('shift', -0.5, -0.5) @ _mat3('scale', 2, 2) \
                @ _mat3('shift', (mx0 - mx1) / 2, (my0 - my1) / 2)
            B = _mat3('scale', 0.5, 0.5) @ _mat3('shift', 0.5, 0.5) @ _mat3('scale', out_w / 2, out_h / 2)
            theta = torch.empty([batch_size, 2, 3], device=device)
            a9 = (ctypes.c_float * 9)(*A.reshape(-1).tolist())
            b9 = (ctypes.c_float * 9)(*B.reshape(-1).tolist())
            with torch.cuda.device(device):
                st = _native.lib().pasta_ada_theta(_native.ptr(G_inv), batch_size, a9, b9, _native.ptr(theta), _native.stream())
            _native.check(st)
            images = grid_sample_gradfix.affine_sample(images, theta, (out_h, out_w))    # no grid tensor; gather adjoint, no atomics
            images = upfirdn2d.downsample2d(x=images, f=self.Hz_geom, down=2, padding=-hz_pad * 2, flip_filter=True)

        if color:           # augment.py:354-364
            if num_channels == 3:
                images = _ColorAffine.apply(images, C, 0)
            elif num_channels == 1:
                row = C[:, :3, :].mean(dim=1, keepdim=True)                     # [N,1,4]
                images = images * row[:, :, :3].sum(dim=2, keepdim=True).unsqueeze(3) + row[:, :, 3:].unsqueeze(3)
            else:
                raise ValueError('Image must be RGB (3 channels) or L (1 channel)')

        u, z = draws['u'], draws['z']
        if self.imgfilter > 0:      # per-sample separable band amplification (augment.py:370-400)
            num_bands = self.Hz_fbank.shape[0]
            assert len(self.imgfilter_bands) == num_bands
            power = torch.as_tensor(np.array([10, 1, 1, 1]) / 13, dtype=torch.float32, device=device)      # expected 1/f spectrum
            u0, z0 = DRAWS_U.index('imgfilter.on.0'), DRAWS_Z.index('imgfilter.0')
            amp = torch.exp2(z[:, z0:z0 + num_bands] * self.imgfilter_std)
            strength = torch.as_tensor(self.imgfilter_bands, dtype=torch.float32, device=device)
            amp = torch.where(u[:, u0:u0 + num_bands] < self.imgfilter * self.p * strength, amp, torch.ones_like(amp))
            if debug_percentile is not None:
                at_percentile = np.exp2(float(torch.erfinv(torch.tensor(debug_percentile * 2 - 1))) * self.imgfilter_std)
                amp = torch.where(strength > 0, torch.full_like(amp, at_percentile), torch.ones_like(amp))
            # band i: gains (1,..,amp_i,..,1), normalised to unit expected power; the global gain is their product
            eye = torch.eye(num_bands, device=device)
            t = 1 + eye * (amp.unsqueeze(2) - 1)                                                            # [N, band i, gains]
            t = t / (power * t.square()).sum(dim=-1, keepdim=True).sqrt()
            g = t.prod(dim=1)                                                                               # [N, bands]
            taps = (g @ self.Hz_fbank).unsqueeze(1).repeat([1, num_channels, 1]).reshape([batch_size * num_channels, 1, -1])
            pad = self.Hz_fbank.shape[1] // 2
            images = images.reshape([1, batch_size * num_channels, height, width])
            images = torch.nn.functional.pad(input=images, pad=[pad, pad, pad, pad], mode='reflect')
            images = conv2d_gradfix.conv2d(input=images, weight=taps.unsqueeze(2), groups=batch_size * num_channels)
            images = conv2d_gradfix.conv2d(input=images, weight=taps.unsqueeze(3), groups=batch_size * num_channels)
            images = images.reshape([batch_size, num_channels, height, width])

        if self.noise > 0:          # additive RGB noise (augment.py:406-412)
            sigma = z[:, DRAWS_Z.index('noise.sigma')].abs() * self.noise_std
            sigma = torch.where(u[:, DRAWS_U.index('noise.on')] < self.noise * self.p, sigma, torch.zeros_like(sigma))
            if debug_percentile is not None:
                sigma = torch.full_like(sigma, float(torch.erfinv(torch.tensor(debug_percentile))) * self.noise_std)
            field = draws['noise_field'] if 'noise_field' in draws else torch.randn([batch_size, num_channels, height, width], device=device)
            images = images + field * sigma.reshape(-1, 1, 1, 1)

        if self.cutout > 0:         # augment.py:414-428
            size = torch.where(u[:, DRAWS_U.index('cutout.on')] < self.cutout * self.p, self.cutout_size, 0.0).to(torch.float32)
            cx, cy = u[:, DRAWS_U.index('cutout.x')], u[:, DRAWS_U.index('cutout.y')]
            if debug_percentile is not None:
                size = torch.full_like(size, self.cutout_size)
                cx = cy = torch.full_like(cx, debug_percentile)
            xs = (torch.arange(width, device=device) + 0.5) / width
            ys = (torch.arange(height, device=device) + 0.5) / height
            keep_x = (xs.reshape(1, 1, 1, -1) - cx.reshape(-1, 1, 1, 1)).abs() >= size.reshape(-1, 1, 1, 1) / 2
            keep_y = (ys.reshape(1, 1, -1, 1) - cy.reshape(-1, 1, 1, 1)).abs() >= size.reshape(-1, 1, 1, 1) / 2
            images = images * torch.logical_or(keep_x, keep_y).to(torch.float32)

        return images

#----------------------------------------------------------------------------
