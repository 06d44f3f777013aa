"""CPU restatement of the PASTA-GAN generator / discriminator -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional form: every function takes a flat ``{name: tensor}`` dict (the ``state_dict`` naming of
the reference's modules, which the product modules share) plus the activations, and evaluates the
layer with the op restatements of ``oracle/ref_ops.py``. No module classes, no caching, CPU only.
Citations are to the reference's ``training/networks.py``.

Parity status: PINNED at ``img_resolution=256`` -- ``oracle/make_golden_models.py`` runs the reference's own
``GeneratorFull`` / ``Discriminator`` on deterministic weights (``oracle/param_fill.py``) and
``tests/test_oracle_golden.py`` checks this file against the stored outputs and gradients.
PARITY UNPINNED for any other ``img_resolution``: the reference ships no generator class for 512x320 (its
``GeneratorFull`` hard-codes 128 / 256: SURVEY F9; test_512.py drives a class that exists only inside an unreleased
pickle).  For those resolutions this file restates THIS REPOSITORY's generalisation (training/networks.py,
``_PatchRoutedSynthesis`` / ``_TryOnGenerator``: pose encoder log2(R) - 2 stages deep, log2(R) - 4 retained-image
feature levels, SPADE stage at R / 2, texture block at R); the tests then check the HIP path against that restatement only.
"""

import numpy as np
import torch
import torch.nn.functional as F

from . import ref_ops as R

F4 = [1, 3, 3, 1]

def _filter():
    return R.setup_filter(F4)

#----------------------------------------------------------------------------
# Layer primitives.

def fc(sd, p, x, activation='linear', lr_multiplier=1.0):
    """FullyConnectedLayer.forward, networks.py:115-128."""
    w = sd[p + '.weight'].to(x.dtype) * (lr_multiplier / np.sqrt(sd[p + '.weight'].shape[1]))
    b = sd.get(p + '.bias')
    if b is not None:
        b = b.to(x.dtype) * lr_multiplier
    if activation == 'linear' and b is not None:
        return x @ w.t() + b[None]
    return R.bias_act(x @ w.t(), b, act=activation)

def conv2d_layer(sd, p, x, activation='linear', up=1, down=1, conv_clamp=None, gain=1):
    """Conv2dLayer.forward, networks.py:170-179."""
    w = sd[p + '.weight']
    k = w.shape[2]
    w = w * (1 / np.sqrt(w.shape[1] * k * k))
    b = sd.get(p + '.bias')
    # 16-bit storage: without upsampling the convolution is the last step and the HIP path applies bias / activation in its
    # epilogue, so nothing is rounded in between
    R.CONV_OUTPUT_ROUNDED = (up != 1)
    try:
        x = R.conv2d_resample(x, w.to(x.dtype), f=_filter(), up=up, down=down, padding=k // 2, flip_weight=(up == 1))
    finally:
        R.CONV_OUTPUT_ROUNDED = True
    act_gain = R.ACT_DEFAULTS[activation][1] * gain
    act_clamp = conv_clamp * gain if conv_clamp is not None else None
    return R.bias_act(x, b.to(x.dtype) if b is not None else None, act=activation, gain=act_gain, clamp=act_clamp)

def spade_conv2d_layer(sd, p, x, activation='relu', conv_clamp=None, gain=1, no_act=False):
    """Spade_Conv2dLayer.forward (activation BEFORE the convolution), networks.py:4342-4355."""
    w = sd[p + '.weight']
    k = w.shape[2]
    w = w * (1 / np.sqrt(w.shape[1] * k * k))
    b = sd.get(p + '.bias')
    if not no_act:
        act_gain = R.ACT_DEFAULTS[activation][1] * gain
        act_clamp = conv_clamp * gain if conv_clamp is not None else None
        x = R.bias_act(x, b, act=activation, gain=act_gain, clamp=act_clamp)
    return R.conv2d_resample(x, w.to(x.dtype), f=_filter(), padding=k // 2, flip_weight=True)

def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """networks.py:30-32"""
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()

def mapping(sd, p, z, c, num_layers, num_ws, z_dim, c_dim, lr_multiplier=0.01):
    """MappingNetwork.forward without the w_avg side effect, networks.py:223-259."""
    x = None
    if z_dim > 0:
        x = normalize_2nd_moment(z.to(torch.float32))
    if c_dim > 0:
        y = normalize_2nd_moment(fc(sd, p + '.embed', c.to(torch.float32)))
        x = torch.cat([x, y], dim=1) if x is not None else y
    for i in range(num_layers):
        x = fc(sd, f'{p}.fc{i}', x, activation='lrelu', lr_multiplier=lr_multiplier)
    if num_ws is not None:
        x = x.unsqueeze(1).repeat([1, num_ws, 1])
    return x

def synthesis_layer(sd, p, x, w, up=1, noise_mode='const', conv_clamp=None, gain=1, fused_modconv=False):
    """SynthesisLayer.forward, networks.py:296-315 (noise_mode 'const' or 'none' only: 'random' draws)."""
    styles = fc(sd, p + '.affine', w)
    noise = None
    if noise_mode == 'const':
        noise = sd[p + '.noise_const'] * sd[p + '.noise_strength']
    weight = sd[p + '.weight']
    x = R.modulated_conv2d(x, weight, styles, noise=noise, up=up, padding=weight.shape[2] // 2, resample_filter=_filter(),
                           flip_weight=(up == 1), fused_modconv=fused_modconv)
    act_gain = R.ACT_DEFAULTS['lrelu'][1] * gain
    act_clamp = conv_clamp * gain if conv_clamp is not None else None
    return R.bias_act(x, sd[p + '.bias'].to(x.dtype), act='lrelu', gain=act_gain, clamp=act_clamp)

def torgb_full(sd, p, x, w, conv_clamp=None, fused_modconv=False):
    """ToRGBLayerFull.forward, networks.py:5600-5611."""
    weight = sd[p + '.weight']
    styles = fc(sd, p + '.affine', w) * (1 / np.sqrt(weight.shape[1] * weight.shape[2] ** 2))
    parsing = None
    if (p + '.m_weight1') in sd:
        parsing = R.modulated_conv2d(x, sd[p + '.m_weight1'], styles, demodulate=False, fused_modconv=fused_modconv)
        parsing = R.bias_act(parsing, sd[p + '.m_bias1'].to(x.dtype), clamp=conv_clamp)
    y = R.modulated_conv2d(x, weight, styles, demodulate=False, fused_modconv=fused_modconv)
    return R.bias_act(y, sd[p + '.bias'].to(x.dtype), clamp=conv_clamp), parsing

#----------------------------------------------------------------------------
# Encoders.

def resblock(sd, p, x, activation='relu', down=1):
    """ResBlock.forward, networks.py:553-558."""
    y = conv2d_layer(sd, p + '.skip', x, down=down, gain=np.sqrt(0.5))
    x = conv2d_layer(sd, p + '.conv0', x, activation=activation, down=down)
    x = conv2d_layer(sd, p + '.conv1', x, activation=activation, gain=np.sqrt(0.5))
    return R.q(y + x)

def const_encoder(sd, p, pose, n_downsampling=6):
    """ConstEncoderNetwork, networks.py:560-579 (n_downsampling = 6 at 256)."""
    x = conv2d_layer(sd, f'{p}.model.0', pose)
    for i in range(n_downsampling):
        x = conv2d_layer(sd, f'{p}.model.{i + 1}', x, down=2)
    return x

def dense(sd, p, x):
    """Dense: per-pixel Linear -> InstanceNorm2d -> LeakyReLU(0.01), networks.py:594-611."""
    out = R.q(F.linear(x.permute(0, 2, 3, 1), R.qw(sd[p + '.linear.weight']), sd[p + '.linear.bias']).permute(0, 3, 1, 2))
    return R.q(F.leaky_relu(R.q(F.instance_norm(out, eps=1e-5)), 0.01))

def style_encoder(sd, p, c, retain, feat_levels=4):
    """StyleEncoderNetworkV16.forward, networks.py:4872-4883 (four feature levels at 256)."""
    feats = []
    x = retain
    for i in range(feat_levels):
        x = conv2d_layer(sd, f'{p}.feat_enc.{i}', x, down=(1 if i == 0 else 2))
        feats.append(x)
    x = conv2d_layer(sd, f'{p}.model.0', c)
    idx = 1
    for stage in range(6):
        x = dense(sd, f'{p}.model.{idx}', x)
        x = conv2d_layer(sd, f'{p}.model.{idx + 1}', x, down=(2 if stage < 3 else 1))
        idx += 2
    x = R.q(x.mean(dim=[2, 3]))
    return fc(sd, p + '.fc', x), feats

#----------------------------------------------------------------------------
# SPADE.

def spade_norm_block(sd, p, x, feat):
    """Spade_Norm_Block.forward, networks.py:4371-4379."""
    normalized = F.instance_norm(x, eps=1e-5)
    actv = torch.relu(spade_conv2d_layer(sd, p + '.conv_mlp', feat, no_act=True))
    gamma = spade_conv2d_layer(sd, p + '.conv_gamma', actv, no_act=True)
    beta = spade_conv2d_layer(sd, p + '.conv_beta', actv, no_act=True)
    return R.q(normalized * (1 + gamma) + beta)

def spade_resblock(sd, p, x, feat, conv_clamp=None):
    """Spade_ResBlockV2.forward, networks.py:5264-5273 (the block is built without conv_clamp)."""
    x = spade_conv2d_layer(sd, p + '.conv', x, no_act=True)
    y = spade_conv2d_layer(sd, p + '.skip', spade_norm_block(sd, p + '.spade_skip', x, feat), gain=np.sqrt(0.5))
    x = spade_conv2d_layer(sd, p + '.conv0', spade_norm_block(sd, p + '.spade0', x, feat))
    x = spade_conv2d_layer(sd, p + '.conv1', spade_norm_block(sd, p + '.spade1', x, feat), gain=np.sqrt(0.5))
    return R.q(y + x)

def get_spade_feat(sd, p, mask_256, denorm_mask, denorm_input, spade_resolution=128):
    """SynthesisNetworkFull.get_spade_feat, networks.py:5777-5800 (the fill count ``128 * 128`` is the SPADE plane)."""
    mask_256 = (mask_256 > 0.9).float()
    mask_128 = (F.interpolate(mask_256, scale_factor=0.5) > 0.9).float()
    denorm_mask_128 = (F.interpolate(denorm_mask, scale_factor=0.5) > 0.9).float()
    valid = ((mask_128 + denorm_mask_128) == 2.0).float()
    res_mask = mask_128 - valid
    x = R.q(denorm_input * mask_256 - (1 - mask_256))
    x = conv2d_layer(sd, p + '.spade_encoder.0', x, activation='relu')
    x = resblock(sd, p + '.spade_encoder.1', x)
    feat = resblock(sd, p + '.spade_encoder.2', x, down=2)
    feat_sum = (feat * valid).sum(dim=(2, 3), keepdim=True)
    mask_sum = valid.sum(dim=(2, 3), keepdim=True)
    ok = (mask_sum > 10).float()
    mask_sum = mask_sum * ok + (spade_resolution * spade_resolution) * (1 - ok)
    if R.STORAGE is None:
        return feat * (1 - res_mask) + (feat_sum / mask_sum) * res_mask
    return R.q(feat * (1 - res_mask)) + R.q((feat_sum / mask_sum) * res_mask)     # two 16-bit tensors, added in 16-bit: exact (disjoint supports)

#----------------------------------------------------------------------------
# Generator.

def synthesis_block_full(sd, p, x, img, ws, pose_feat, cat_feat, res, first, conv_clamp, noise_mode, fused_modconv):
    """SynthesisBlockFull.forward ('skip' architecture, fp32), networks.py:5669-5719."""
    wi = iter(ws.unbind(dim=1))
    if first:
        x = synthesis_layer(sd, p + '.conv1', pose_feat, next(wi), conv_clamp=conv_clamp, noise_mode=noise_mode, fused_modconv=fused_modconv)
    else:
        x = synthesis_layer(sd, p + '.conv0', x, next(wi), up=2, conv_clamp=conv_clamp, noise_mode=noise_mode, fused_modconv=fused_modconv)
        x = synthesis_layer(sd, p + '.conv1', x, next(wi), conv_clamp=conv_clamp, noise_mode=noise_mode, fused_modconv=fused_modconv)
        if x.shape[2] > 16:
            x = conv2d_layer(sd, p + '.merge_conv', torch.cat([x, cat_feat[str(x.shape[2])]], dim=1))
    if img is not None:
        with R.fp32_region():           # the running image is fp32 in every storage mode (networks.py:5713-5716)
            img = R.upsample2d(img, _filter())
    y, parsing = torgb_full(sd, p + '.torgb', x, next(wi), conv_clamp=conv_clamp, fused_modconv=fused_modconv)
    img = img + y if img is not None else y
    return x, img, parsing

def synthesis_full(sd, p, ws, pose_feat, cat_feat, du_in, dl_in, du_mask, dl_mask, img_resolution=256, conv_clamp=256,
                   noise_mode='const', fused_modconv=False):
    """SynthesisNetworkFull.forward, networks.py:5803-5840."""
    resolutions = [2 ** i for i in range(2, int(np.log2(img_resolution)) + 1)]
    x = img = parsing = None
    w_idx = 0
    block_ws = []
    for res in resolutions:
        nconv = 1 if res == 4 else 2
        block_ws.append(ws.narrow(1, w_idx, nconv + 1))
        w_idx += nconv
    for res, cur in zip(resolutions, block_ws):
        x, img, parsing = synthesis_block_full(sd, f'{p}.b{res}', x, img, cur, pose_feat, cat_feat, res, res == 4, conv_clamp,
                                               noise_mode, fused_modconv)
        if res == img_resolution // 2:
            x_128, img_128 = x, img
    sres = img_resolution // 2
    index = torch.argmax(torch.softmax(parsing.detach(), dim=1), dim=1)[:, None].float()
    upper = get_spade_feat(sd, p, (index == 1).float(), du_mask, du_in, sres)
    lower = get_spade_feat(sd, p, (index == 2).float(), dl_mask, dl_in, sres)
    feat = torch.cat([upper, lower], dim=1)
    xs = x_128
    for i in (1, 2, 3):
        xs = spade_resblock(sd, f'{p}.spade_b{sres}_{i}', xs, feat)
    _, finetune, _ = synthesis_block_full(sd, f'{p}.texture_b{img_resolution}', xs, img_128, block_ws[-1], pose_feat, cat_feat,
                                          img_resolution, False, conv_clamp, noise_mode, fused_modconv)
    return img, finetune, parsing

def generator_full(sd, z, c, retain, pose, du_in, dl_in, du_mask, dl_mask, img_resolution=256, conv_clamp=256,
                   mapping_layers=1, noise_mode='const', fused_modconv=False):
    """GeneratorFull.forward, networks.py:5866-5881."""
    log2 = int(np.log2(img_resolution))
    pose, c, retain = R.q(pose), R.q(c), R.q(retain)       # 16-bit storage: the encoders' inputs are stored in it as well
    pose_feat = const_encoder(sd, 'const_encoding', pose, n_downsampling=log2 - 2)
    code, feats = style_encoder(sd, 'style_encoding', c, retain, feat_levels=log2 - 4)
    num_ws = 2 * int(np.log2(img_resolution)) - 2      # 1 + 2*(blocks-1) convs + the last ToRGB
    ws = mapping(sd, 'mapping', z, code, mapping_layers, num_ws, z_dim=(z.shape[1] if z is not None else 0), c_dim=code.shape[1])
    cat = {str(f.shape[2]): f for f in feats}
    return synthesis_full(sd, 'synthesis', ws, pose_feat, cat, du_in, dl_in, du_mask, dl_mask, img_resolution, conv_clamp,
                          noise_mode, fused_modconv)

#----------------------------------------------------------------------------
# GeneratorV18 (the released 256 inference model of test.py).

def torgb_v18(sd, p, x, w, conv_clamp=None, fused_modconv=False):
    """ToRGBLayerV18.forward, networks.py:5296-5310."""
    weight = sd[p + '.weight']
    styles = fc(sd, p + '.affine', w) * (1 / np.sqrt(weight.shape[1] * weight.shape[2] ** 2))
    masks = [None, None]
    if (p + '.m_weight1') in sd:
        for i, tag in enumerate(['1', '2']):
            m = R.modulated_conv2d(x, sd[p + '.m_weight' + tag], styles, demodulate=False, fused_modconv=fused_modconv)
            masks[i] = R.bias_act(m, sd[p + '.m_bias' + tag].to(x.dtype), clamp=conv_clamp, act='sigmoid')
    y = R.modulated_conv2d(x, weight, styles, demodulate=False, fused_modconv=fused_modconv)
    return R.bias_act(y, sd[p + '.bias'].to(x.dtype), clamp=conv_clamp), masks[0], masks[1]

def _block_v18(sd, p, x, img, ws, pose_feat, cat_feat, first, conv_clamp, noise_mode, fused_modconv):
    """SynthesisBlockV18.forward, networks.py:5369-5418."""
    wi = iter(ws.unbind(dim=1))
    if first:
        x = synthesis_layer(sd, p + '.conv1', pose_feat, next(wi), conv_clamp=conv_clamp, noise_mode=noise_mode, fused_modconv=fused_modconv)
    else:
        x = synthesis_layer(sd, p + '.conv0', x, next(wi), up=2, conv_clamp=conv_clamp, noise_mode=noise_mode, fused_modconv=fused_modconv)
        x = synthesis_layer(sd, p + '.conv1', x, next(wi), conv_clamp=conv_clamp, noise_mode=noise_mode, fused_modconv=fused_modconv)
        if x.shape[2] > 16:
            x = conv2d_layer(sd, p + '.merge_conv', torch.cat([x, cat_feat[str(x.shape[2])]], dim=1))
    if img is not None:
        with R.fp32_region():           # the running image is fp32 in every storage mode (networks.py:5713-5716)
            img = R.upsample2d(img, _filter())
    y, um, lm = torgb_v18(sd, p + '.torgb', x, next(wi), conv_clamp=conv_clamp, fused_modconv=fused_modconv)
    img = img + y if img is not None else y
    return x, img, um, lm

def generator_v18(sd, z, c, retain, pose, du_in, dl_in, du_mask, dl_mask, img_resolution=256, conv_clamp=256,
                  mapping_layers=1, noise_mode='const', fused_modconv=True):
    """GeneratorV18.forward + SynthesisNetworkV18.forward, networks.py:5481-5577."""
    log2 = int(np.log2(img_resolution))
    pose_feat = const_encoder(sd, 'const_encoding', pose, n_downsampling=log2 - 2)
    code, feats = style_encoder(sd, 'style_encoding', c, retain, feat_levels=log2 - 4)
    num_ws = 2 * int(np.log2(img_resolution)) - 2
    ws = mapping(sd, 'mapping', z, code, mapping_layers, num_ws, z_dim=(z.shape[1] if z is not None else 0), c_dim=code.shape[1])
    cat = {str(f.shape[2]): f for f in feats}
    resolutions = [2 ** i for i in range(2, int(np.log2(img_resolution)) + 1)]
    x = img = um = lm = None
    w_idx, block_ws = 0, []
    for res in resolutions:
        nconv = 1 if res == 4 else 2
        block_ws.append(ws.narrow(1, w_idx, nconv + 1))
        w_idx += nconv
    for res, cur in zip(resolutions, block_ws):
        x, img, um, lm = _block_v18(sd, f'synthesis.b{res}', x, img, cur, pose_feat, cat, res == 4, conv_clamp, noise_mode, fused_modconv)
        if res == img_resolution // 2:
            x_128, img_128 = x, img
    sres = img_resolution // 2
    feat = torch.cat([get_spade_feat(sd, 'synthesis', um.detach(), du_mask, du_in, sres),
                      get_spade_feat(sd, 'synthesis', lm.detach(), dl_mask, dl_in, sres)], dim=1)
    xs = x_128
    for i in (1, 2, 3):
        xs = spade_resblock(sd, f'synthesis.spade_b{sres}_{i}', xs, feat)
    _, finetune, _, _ = _block_v18(sd, f'synthesis.texture_b{img_resolution}', xs, img_128, block_ws[-1], pose_feat, cat, False, conv_clamp,
                                   noise_mode, fused_modconv)
    return img, finetune, um, lm

#----------------------------------------------------------------------------
# Discriminator.

def minibatch_std(x, group_size=4, num_channels=1):
    """MinibatchStdLayer.forward, networks.py:1007-1022."""
    N, C, H, W = x.shape
    G = min(group_size, N)
    Fc = num_channels
    y = x.reshape(G, -1, Fc, C // Fc, H, W)
    y = y - y.mean(dim=0)
    y = (y.square().mean(dim=0) + 1e-8).sqrt()
    y = y.mean(dim=[2, 3, 4]).reshape(-1, Fc, 1, 1).repeat(G, 1, H, W)
    return torch.cat([x, y], dim=1)

def discriminator(sd, img, c, img_resolution=256, conv_clamp=256, mapping_layers=8):
    """Discriminator.forward ('resnet' architecture, fp32), networks.py:1128-1139 with blocks :973-996 and epilogue :1057-1080."""
    x = None
    for res in [2 ** i for i in range(int(np.log2(img_resolution)), 2, -1)]:
        p = f'b{res}'
        if res == img_resolution:
            x = conv2d_layer(sd, p + '.fromrgb', R.q(img), activation='lrelu', conv_clamp=conv_clamp)
        y = conv2d_layer(sd, p + '.skip', x, down=2, gain=np.sqrt(0.5))
        x = conv2d_layer(sd, p + '.conv0', x, activation='lrelu', conv_clamp=conv_clamp)
        x = conv2d_layer(sd, p + '.conv1', x, activation='lrelu', down=2, conv_clamp=conv_clamp, gain=np.sqrt(0.5))
        x = R.q(y + x)
    with R.fp32_region():               # the epilogue always computes in fp32 (networks.py:1060)
        cmap = mapping(sd, 'mapping', None, c, mapping_layers, None, z_dim=0, c_dim=c.shape[1])
        x = minibatch_std(x)
        x = conv2d_layer(sd, 'b4.conv', x, activation='lrelu', conv_clamp=conv_clamp)
        x = fc(sd, 'b4.fc', x.flatten(1), activation='lrelu')
        x = fc(sd, 'b4.out', x)
        return (x * cmap).sum(dim=1, keepdim=True) * (1 / np.sqrt(cmap.shape[1]))

#----------------------------------------------------------------------------
