"""Loss and gradient accumulation for the full-body try-on generator / discriminator.

Mirrors ``StyleGAN2Loss`` of the reference's training/loss_wo_flow_fullbody.py (constructor :33-72,
``run_G`` :74-94, ``run_D`` :96-102, ``accumulate_gradients`` :106-254): non-saturating GAN terms on
both generator outputs, L1, 6-class parsing cross-entropy and lazy R1, with DistributedDataParallel
synchronisation gated exactly as there. Differences, all outside the G/D kernels:
the VGG19 perceptual term (``VGGLoss`` :259-273 over ``VGG19_Feature`` :275-310) runs on this package's convolution
when ``vgg_weight > 0``; its pretrained weights (``./checkpoints/vgg19-dcbb9e9d.pth``) cannot be obtained here, so the
file is loaded when present and otherwise ``vgg_random_init=True`` must be passed (random weights: the arithmetic and
its cost are the reference's, the loss value is not meaningful).  The contextual term needs a second unobtainable
checkpoint and is rejected unless its weight is 0; statistics reporting is a no-op hook; style mixing picks its cutoff
without a host synchronisation.
"""

import os

import numpy as np
import torch

from torch_utils import misc
from torch_utils.ops import conv2d_gradfix

#----------------------------------------------------------------------------

# torchvision's configuration 'E' (VGG-19 without batch norm), as listed in the reference (:385); 'M' = 2x2 max pooling.
VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
# Feature taps of VGG19_Feature (:293-302): outputs of layers 1, 6, 11, 20, 29 of ``features`` = relu1_1, relu2_1, relu3_1,
# relu4_1, relu5_1, i.e. after convolution number 1, 3, 5, 9, 13 (1-based).
VGG19_TAPS = (1, 3, 5, 9, 13)

class VGG19_Feature(torch.nn.Module):
    """relu{1..5}_1 activations of VGG-19 (:275-310).  Weights are frozen buffers named as torchvision's
    ``features.<index>.weight / .bias`` so that the reference's checkpoint loads unchanged."""
    def __init__(self, device, ckpt_path='./checkpoints/vgg19-dcbb9e9d.pth', random_init=False, seed=0):
        super().__init__()
        self.layers = []                                    # ('conv', features-index) / ('pool', None), up to relu5_1
        idx, cin, convs = 0, 3, 0
        gen = torch.Generator().manual_seed(seed)
        state = None
        if os.path.isfile(ckpt_path):
            state = torch.load(ckpt_path, map_location='cpu')
        elif not random_init:
            raise FileNotFoundError(f'{ckpt_path} not found: pass vgg_random_init=True to run the perceptual term with random weights')
        for v in VGG19_CFG:
            if v == 'M':
                self.layers.append(('pool', None)); idx += 1
                continue
            if state is not None:
                w, b = state[f'features.{idx}.weight'], state[f'features.{idx}.bias']
            else:       # He-normal, like torchvision's initialisation of an untrained VGG
                w = torch.randn([v, cin, 3, 3], generator=gen) * (2.0 / (9 * v)) ** 0.5
                b = torch.zeros([v])
            self.register_buffer(f'features_{idx}_weight', w.float().to(device))
            self.register_buffer(f'features_{idx}_bias', b.float().to(device))
            self.layers.append(('conv', idx)); idx += 2; cin = v; convs += 1
            if convs == VGG19_TAPS[-1]:
                break

    def forward(self, x):
        feats, convs = [], 0
        for kind, idx in self.layers:
            if kind == 'pool':
                x = torch.nn.functional.max_pool2d(x, kernel_size=2, stride=2)
                continue
            x = conv2d_gradfix.conv2d_bias_act(x, getattr(self, f'features_{idx}_weight'), getattr(self, f'features_{idx}_bias'),
                                               padding=1, act='relu', gain=1)
            convs += 1
            if convs in VGG19_TAPS:
                feats.append(x)
        return feats

class VGGLoss(torch.nn.Module):
    """sum_i w_i * L1(vgg_i(x), vgg_i(y).detach()) with w = 1/32, 1/16, 1/8, 1/4, 1 (:259-273)."""
    def __init__(self, device, random_init=False, weights=(1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)):
        super().__init__()
        self.vgg = VGG19_Feature(device, random_init=random_init)
        self.weights = weights

    def forward(self, x, y):
        x_vgg, y_vgg = self.vgg(x), self.vgg(y)
        loss = 0
        for w, fx, fy in zip(self.weights, x_vgg, y_vgg):
            loss = loss + w * torch.nn.functional.l1_loss(fx, fy.detach())
        return loss

#----------------------------------------------------------------------------

class Loss:
    def accumulate_gradients(self, phase, real_img, gen_z, style_input, retain, pose, denorm_upper_input, denorm_lower_input,
                             denorm_upper_mask, denorm_lower_mask, gt_parsing, sync, gain): # to be overridden by subclass
        raise NotImplementedError()

#----------------------------------------------------------------------------

class StyleGAN2Loss(Loss):
    def __init__(self, device, G_mapping, G_synthesis, G_const_encoding, G_style_encoding, D, augment_pipe=None,
                 style_mixing_prob=0.9, r1_gamma=10, pl_batch_shrink=2, pl_decay=0.01, pl_weight=0, l1_weight=50,
                 vgg_weight=50, contextual_weight=1.0, mask_weight=1.0, report_fn=None, vgg_random_init=False):
        super().__init__()
        if contextual_weight > 0:
            raise NotImplementedError('the contextual term needs ./checkpoints/vgg19_conv.pth, which cannot be obtained here; '
                                      'pass contextual_weight=0 (accumulate_gradients never uses it: :106-254)')
        if pl_weight != 0:
            raise NotImplementedError('path-length regularisation is unreachable in the reference (wrong arity at '
                                      'loss_wo_flow_fullbody.py:185-205; train.sh sets --pl_weight 0)')
        self.device = device
        self.G_mapping = G_mapping
        self.G_synthesis = G_synthesis
        self.G_const_encoding = G_const_encoding
        self.G_style_encoding = G_style_encoding
        self.D = D
        self.augment_pipe = augment_pipe
        self.style_mixing_prob = style_mixing_prob
        self.r1_gamma = r1_gamma
        self.pl_weight = pl_weight
        self.l1_weight = l1_weight
        self.vgg_weight = vgg_weight
        self.mask_weight = mask_weight
        self.report = report_fn if report_fn is not None else (lambda name, value: None)
        if vgg_weight > 0:
            self.criterionVGG = VGGLoss(device, random_init=vgg_random_init)
        class_weight = torch.tensor([1, 2, 2, 3, 3, 3], dtype=torch.float32, device=device)
        self.ce_parsing = torch.nn.CrossEntropyLoss(ignore_index=255, weight=class_weight)

    def run_G(self, z, c, pose, const_feats, denorm_upper_mask, denorm_lower_mask, denorm_upper_input, denorm_lower_input, sync):
        cat_feats = {str(f.shape[2]): f for f in const_feats}
        with misc.ddp_sync(self.G_const_encoding, sync):
            pose_feat = self.G_const_encoding(pose)
        with misc.ddp_sync(self.G_mapping, sync):
            ws = self.G_mapping(z, c)
            if self.style_mixing_prob > 0:
                num_ws = ws.shape[1]
                cutoff = torch.empty([], dtype=torch.int64, device=ws.device).random_(1, num_ws)
                cutoff = torch.where(torch.rand([], device=ws.device) < self.style_mixing_prob, cutoff, torch.full_like(cutoff, num_ws))
                ws2 = self.G_mapping(torch.randn_like(z), c, skip_w_avg_update=True)
                keep = (torch.arange(num_ws, device=ws.device) < cutoff).reshape(1, num_ws, 1)
                ws = torch.where(keep, ws, ws2)          # ws[:, cutoff:] = ws2[:, cutoff:] without reading cutoff on the host
        with misc.ddp_sync(self.G_synthesis, sync):
            img, finetune_img, pred_parsing = self.G_synthesis(ws, pose_feat, cat_feats, denorm_upper_input, denorm_lower_input,
                                                               denorm_upper_mask, denorm_lower_mask)
        return img, finetune_img, pred_parsing, ws

    def run_D(self, img, c, sync):
        if self.augment_pipe is not None:
            img = self.augment_pipe(img)
        with misc.ddp_sync(self.D, sync):
            logits = self.D(img, c)
        return logits

    def _mbstd_groups(self, n):
        """Number of minibatch-std groups a batch of n splits into (networks.py:1007-1022), or None when k such batches
        cannot share one discriminator pass with unchanged statistics: a layer without a group size (its group is the
        whole batch), a group size above n (the merged batch of k*n would form larger groups than n does), or n not a
        multiple of the group size."""
        D = self.D.module if hasattr(self.D, 'module') else self.D
        sizes = {m.group_size for m in D.modules() if type(m).__name__ == 'MinibatchStdLayer'}
        if not sizes:
            return n                      # nothing couples the samples: any stacking order works
        if None in sizes or len(sizes) != 1:
            return None
        g = int(next(iter(sizes)))
        return n // g if 0 < g <= n and n % g == 0 else None

    @staticmethod
    def backward_passes(phase):
        """How many ``backward()`` calls of one ``accumulate_gradients(phase)`` reach the phase's module (the gradient
        reducer overlaps its all-reduce with the last of them): Gmain / Dmain / Dreg one each, Dboth two (:227, :254),
        Greg none (pl_weight = 0: only the style encoder's forward runs)."""
        return {'Gmain': 1, 'Gboth': 1, 'Greg': 0, 'Dmain': 1, 'Dreg': 1, 'Dboth': 2}[phase]

    def run_D_multi(self, imgs, cs, sync):
        """``[run_D(img, c) for img, c in zip(imgs, cs)]`` in ONE discriminator pass (the reference calls the
        discriminator once per image batch, :127-128, :214-215, :235; its 4..32-pixel layers are far too small to fill the
        chip at batch 16: measured in the full step, Gmain 132.6 -> 130.9 ms and Dmain 78.9 -> 74.7 ms).
        Everything in D is per sample except the minibatch standard deviation, whose groups are the samples
        ``{m, m + B, m + 2B, ...}`` of a batch split as [G, B] (B groups).  Stacking the batches as [G, k, B] keeps
        every group inside its own batch, so the logits equal those of the separate calls."""
        n = imgs[0].shape[0]
        B = self._mbstd_groups(n) if len(imgs) > 1 and all(i.shape == imgs[0].shape for i in imgs) else None
        if B is None:
            return [self.run_D(img, c, sync) for img, c in zip(imgs, cs)]
        if self.augment_pipe is not None:       # per-sample independent transforms: one call for all k batches
            imgs = list(self.augment_pipe(torch.cat(imgs)).split(n))
        k, G = len(imgs), n // B
        def merge(ts):
            return torch.stack([t.reshape(G, B, *t.shape[1:]) for t in ts], dim=1).reshape(k * n, *ts[0].shape[1:])
        with misc.ddp_sync(self.D, sync):
            logits = self.D(merge(imgs), merge(cs))
        return [t.reshape(n, *logits.shape[1:]) for t in logits.reshape(G, k, B, *logits.shape[1:]).unbind(dim=1)]

    def accumulate_gradients(self, phase, real_img, gen_z, style_input, retain, pose, denorm_upper_input, denorm_lower_input,
                             denorm_upper_mask, denorm_lower_mask, gt_parsing, sync, gain):
        assert phase in ['Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth']
        do_Gmain = (phase in ['Gmain', 'Gboth'])
        do_Dmain = (phase in ['Dmain', 'Dboth'])
        do_Dr1   = (phase in ['Dreg', 'Dboth']) and (self.r1_gamma != 0)
        softplus = torch.nn.functional.softplus

        # The style code doubles as the conditioning label of both real and generated images (:114-116).
        # Only Gmain back-propagates into the encoder. The reference passes `sync` here in every phase
        # (:114); a DDP forward that announces a gradient reduction which never happens leaves the reducer
        # armed and corrupts the next no_sync round on current PyTorch, so the other phases suppress it.
        with misc.ddp_sync(self.G_style_encoding, sync and do_Gmain):
            real_c, cat_feats = self.G_style_encoding(style_input, retain)
            gen_c = real_c
        g_args = (denorm_upper_mask, denorm_lower_mask, denorm_upper_input, denorm_lower_input)

        # Gmain: maximise logits for both generated images, plus reconstruction terms (:119-182).
        if do_Gmain:
            gen_img, gen_finetune_img, pred_parsing, _ws = self.run_G(gen_z, gen_c, pose, cat_feats, *g_args, sync=sync)
            gen_logits, gen_finetune_logits = self.run_D_multi([gen_img, gen_finetune_img], [gen_c, gen_c], sync=False)
            self.report('Loss/scores/fake', gen_logits)
            loss_Gmain = softplus(-gen_logits).mean()
            loss_Gmain_finetune = softplus(-gen_finetune_logits).mean()
            loss_G_L1 = loss_G_finetune_L1 = 0
            if self.l1_weight > 0:
                loss_G_L1 = torch.nn.functional.l1_loss(gen_img, real_img) * self.l1_weight
                loss_G_finetune_L1 = torch.nn.functional.l1_loss(gen_finetune_img, real_img) * self.l1_weight
            loss_mask = 0
            if self.mask_weight > 0:
                loss_mask = torch.mean(self.ce_parsing(pred_parsing, gt_parsing.long()[:, 0, ...])) * self.mask_weight
            loss_G_VGG = loss_G_finetune_VGG = 0
            if self.vgg_weight > 0:       # :162-171
                loss_G_VGG = self.criterionVGG(gen_img, real_img) * self.vgg_weight
                loss_G_finetune_VGG = self.criterionVGG(gen_finetune_img, real_img) * self.vgg_weight
            self.report('Loss/G/vgg', loss_G_VGG)
            self.report('Loss/G/vgg_finetune', loss_G_finetune_VGG)
            loss_G = (loss_Gmain + loss_Gmain_finetune) / 2 + (loss_G_L1 + loss_G_finetune_L1) / 2 + \
                     (loss_G_VGG + loss_G_finetune_VGG) / 2 + loss_mask
            self.report('Loss/scores/fake_finetune', gen_finetune_logits)
            self.report('Loss/G/loss', loss_Gmain)
            self.report('Loss/G/loss_finetune', loss_Gmain_finetune)
            self.report('Loss/G/L1', loss_G_L1)
            self.report('Loss/G/L1_finetune', loss_G_finetune_L1)
            self.report('Loss/G/mask_loss', loss_mask)
            loss_G.mul(gain).backward()

        # Dmain without R1: generated, fine-tuned and real images in one discriminator pass; the reference's two backward
        # calls (:227, :254) add up to the gradient of the summed loss.
        if do_Dmain and not do_Dr1:
            gen_img, gen_finetune_img, _, _ws = self.run_G(gen_z, gen_c, pose, cat_feats, *g_args, sync=False)
            gen_logits, gen_finetune_logits, real_logits = self.run_D_multi(
                [gen_img, gen_finetune_img, real_img.detach()], [gen_c, gen_c, real_c], sync=sync)
            self.report('Loss/scores/fake', gen_logits)
            self.report('Loss/scores/fake_finetune', gen_finetune_logits)
            self.report('Loss/scores/real', real_logits)
            self.report('Loss/signs/real', real_logits.sign())
            loss_Dgen, loss_Dgen_finetune, loss_Dreal = softplus(gen_logits), softplus(gen_finetune_logits), softplus(-real_logits)
            self.report('Loss/D/loss', loss_Dgen + loss_Dreal)
            ((loss_Dgen.mean() + loss_Dgen_finetune.mean()) / 2 + loss_Dreal.mean()).mul(gain).backward()
            return

        # Dmain: minimise logits for generated images (:210-228).
        loss_Dgen = 0
        if do_Dmain:
            gen_img, gen_finetune_img, _, _ws = self.run_G(gen_z, gen_c, pose, cat_feats, *g_args, sync=False)
            gen_logits, gen_finetune_logits = self.run_D_multi([gen_img, gen_finetune_img], [gen_c, gen_c], sync=False)   # gets synced by loss_Dreal
            loss_Dgen = softplus(gen_logits)
            loss_Dgen_finetune = softplus(gen_finetune_logits)
            ((loss_Dgen.mean() + loss_Dgen_finetune.mean()) / 2).mul(gain).backward()

        # Dmain: maximise logits for real images.  Dr1: R1 penalty on real images (:232-254).
        if do_Dmain or do_Dr1:
            real_img_tmp = real_img.detach().requires_grad_(do_Dr1)
            real_logits = self.run_D(real_img_tmp, real_c, sync=sync)
            self.report('Loss/scores/real', real_logits)
            self.report('Loss/signs/real', real_logits.sign())
            loss_Dreal = 0
            if do_Dmain:
                loss_Dreal = softplus(-real_logits)
                self.report('Loss/D/loss', loss_Dgen + loss_Dreal)
            loss_Dr1 = 0
            if do_Dr1:
                with conv2d_gradfix.no_weight_gradients():
                    r1_grads = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_img_tmp], create_graph=True, only_inputs=True)[0]
                r1_penalty = r1_grads.square().sum([1, 2, 3])
                loss_Dr1 = r1_penalty * (self.r1_gamma / 2)
                self.report('Loss/r1_penalty', r1_penalty)
                self.report('Loss/D/reg', loss_Dr1)
            (real_logits * 0 + loss_Dreal + loss_Dr1).mean().mul(gain).backward()

#----------------------------------------------------------------------------
