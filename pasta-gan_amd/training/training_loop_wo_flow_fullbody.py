"""The data-parallel training step of PASTA-GAN's full-body model.

Mirrors the hot loop of the reference's training/training_loop_wo_flow_fullbody.py: module
construction (:274-277), replica synchronisation and gradient exchange (:312-324), lazily regularised Adam
phases Gmain/Greg/Dmain/Dreg (:332-349), gradient accumulation with ``sync`` only on the last round
(:484-505), gradient ``nan_to_num`` + optimiser step (:508-516) and the generator EMA (:521-529).
One process per GPU; gradients are all-reduced by RCCL (``backend='nccl'`` on ROCm) over xGMI, overlapped
with backward: by default through one ``FlatGradReducer`` per optimised module (``ddp_mode='flat'``,
training/grad_reducer.py), or through the reference's five DistributedDataParallel wrappers (``ddp_mode='torch'``).

Dataset loading, snapshots, image grids, ADA and metrics of the reference loop are host
orchestration outside this path; ``SyntheticFullBodyBatch`` supplies tensors of the dataset's
shapes (training_loop...:289-297, 425-456) directly in HBM.
"""

import copy

import numpy as np
import os
import torch

import dnnlib
from torch_utils import misc
from training.grad_reducer import FlatGradReducer, broadcast_module_states

#----------------------------------------------------------------------------

AUGPIPE_SPECS = {    # train_wo_flow_fullbody.py:297-309: which AugmentPipe multipliers each --augpipe name switches on
    'blit':   ['xflip', 'rotate90', 'xint'],
    'geom':   ['scale', 'rotate', 'aniso', 'xfrac'],
    'color':  ['brightness', 'contrast', 'lumaflip', 'hue', 'saturation'],
    'filter': ['imgfilter'],
    'noise':  ['noise'],
    'cutout': ['cutout'],
}
for _name, _parts in [('bg', ['blit', 'geom']), ('bgc', ['blit', 'geom', 'color']), ('bgcf', ['blit', 'geom', 'color', 'filter']),
                      ('bgcfn', ['blit', 'geom', 'color', 'filter', 'noise']), ('bgcfnc', ['blit', 'geom', 'color', 'filter', 'noise', 'cutout'])]:
    AUGPIPE_SPECS[_name] = [m for part in _parts for m in AUGPIPE_SPECS[part]]

def augment_options(aug='ada', augpipe='bgc', p=None, target=None):
    """The --aug / --p / --target / --augpipe options (train_wo_flow_fullbody.py:249-313) as TrainingStep config entries:
    'ada' adapts p towards ``target`` (default 0.6) from ``p`` (default 0); 'fixed' keeps ``p``; 'noaug' -> {}."""
    if aug == 'noaug':
        return dnnlib.EasyDict()
    assert aug in ('ada', 'fixed') and augpipe in AUGPIPE_SPECS
    assert aug != 'fixed' or p is not None, '--aug=fixed requires p'
    opts = dnnlib.EasyDict(augment_kwargs=dnnlib.EasyDict(class_name='training.augment.AugmentPipe', **{m: 1 for m in AUGPIPE_SPECS[augpipe]}),
                           augment_p=float(p) if p is not None else 0.0)
    if aug == 'ada':
        opts.ada_target = 0.6 if target is None else float(target)
    return opts

def fashion_config(channel_base=16384, d_fp16_res=0, mbstd_group_size=4, img_resolution=256, act_dtype=None):
    """G/D/optimiser/loss options of ``--cfg fashion`` (train_wo_flow_fullbody.py:166-215, train.sh:3-10); augmentation
    off (``cfg.update(augment_options(...))`` turns it on).  ``act_dtype`` ('bfloat16' / 'float16'; BASELINE config 5):
    16-bit activation storage in the generator's synthesis network and encoders and in every discriminator block."""
    G_kwargs = dnnlib.EasyDict(class_name='training.networks.GeneratorFull', z_dim=0, c_dim=512, w_dim=512, img_resolution=img_resolution,
                               img_channels=3, mapping_kwargs=dnnlib.EasyDict(num_layers=1),
                               synthesis_kwargs=dnnlib.EasyDict(channel_base=channel_base, channel_max=512, num_fp16_res=3,
                                                                conv_clamp=256, use_noise=True))
    D_kwargs = dnnlib.EasyDict(class_name='training.networks.Discriminator', c_dim=512, img_resolution=img_resolution, img_channels=3,
                               channel_base=channel_base, channel_max=512, num_fp16_res=d_fp16_res, conv_clamp=256,
                               block_kwargs=dnnlib.EasyDict(), mapping_kwargs=dnnlib.EasyDict(),
                               epilogue_kwargs=dnnlib.EasyDict(mbstd_group_size=mbstd_group_size))
    if act_dtype is not None:
        G_kwargs.synthesis_kwargs.act_dtype = act_dtype
        D_kwargs.half_dtype = act_dtype
        D_kwargs.num_fp16_res = int(np.log2(img_resolution)) - 2      # every block b<R> .. b8
    opt = dnnlib.EasyDict(class_name='torch.optim.Adam', lr=0.002, betas=[0, 0.99], eps=1e-8)
    loss_kwargs = dnnlib.EasyDict(class_name='training.loss_wo_flow_fullbody.StyleGAN2Loss', r1_gamma=10, l1_weight=40,
                                  vgg_weight=0, contextual_weight=0, pl_weight=0, mask_weight=20)
    return dnnlib.EasyDict(G_kwargs=G_kwargs, D_kwargs=D_kwargs, G_opt_kwargs=opt, D_opt_kwargs=dnnlib.EasyDict(opt),
                           loss_kwargs=loss_kwargs, ema_kimg=10, ema_rampup=None, G_reg_interval=4, D_reg_interval=16)

#----------------------------------------------------------------------------

class SyntheticFullBodyBatch:
    """Device-resident synthetic batch with the dataset's tensor shapes and value ranges."""
    KEYS = ['real_img', 'style_input', 'retain', 'pose', 'denorm_upper_input', 'denorm_lower_input',
            'denorm_upper_mask', 'denorm_lower_mask', 'gt_parsing']

    def __init__(self, batch, device, seed=0, res=256):
        g = torch.Generator(device='cpu').manual_seed(1234 + seed)
        n = batch
        def u(*shape):
            return torch.rand(shape, generator=g) * 2 - 1
        def blobs(p):
            coarse = torch.rand([n, 1, res // 16, res // 16], generator=g)
            return (torch.nn.functional.interpolate(coarse, size=(res, res), mode='bilinear', align_corners=False) < p).float()
        real_img = u(n, 3, res, res)
        real_img[..., : res // 8] = 1.0               # 192-wide content, white padded to a square (dataset.py:520-524)
        real_img[..., res - res // 8:] = 1.0
        mask = blobs(0.5)
        retain = mask * real_img - (1 - mask)
        lines = (torch.rand([n, 3, res, res], generator=g) < 0.02).float() * 2 - 1
        style = u(n, 42, res // 4, res // 4)
        drop = (torch.rand([n, 14, 1, 1], generator=g) < 0.3).repeat_interleave(3, dim=1)
        style = torch.where(drop, -torch.ones_like(style), style)
        du_mask, dl_mask = blobs(0.35), blobs(0.35)
        gt = torch.randint(0, 6, [n, 1, res // 8, res // 8], generator=g).float()
        t = dict(real_img=real_img, style_input=style, retain=retain, pose=torch.cat([lines, retain], dim=1),
                 denorm_upper_input=u(n, 3, res, res) * du_mask - (1 - du_mask),
                 denorm_lower_input=u(n, 3, res, res) * dl_mask - (1 - dl_mask),
                 denorm_upper_mask=du_mask, denorm_lower_mask=dl_mask,
                 gt_parsing=torch.nn.functional.interpolate(gt, size=(res, res), mode='nearest'))
        self.tensors = {k: v.to(device) for k, v in t.items()}
        self.batch = n

    def split(self, batch_gpu):
        parts = {k: v.split(batch_gpu) for k, v in self.tensors.items()}
        return [{k: parts[k][i] for k in self.KEYS} for i in range(len(parts['real_img']))]

#----------------------------------------------------------------------------

class TrainingStep:
    """Owns G, D, G_ema, the DDP wrappers, the loss and the four optimiser phases; ``run()`` executes one
    iteration of the reference's hot loop on a device-resident batch."""

    def __init__(self, device, cfg=None, num_gpus=1, rank=0, batch_size=16, batch_gpu=16, random_seed=0, ddp_bucket_mb=None, ddp_mode='flat'):
        cfg = cfg if cfg is not None else fashion_config(mbstd_group_size=min(batch_gpu, 4))
        assert ddp_mode in ('flat', 'torch')
        self.device, self.num_gpus, self.rank = device, num_gpus, rank
        self.batch_size, self.batch_gpu = batch_size, batch_gpu
        assert batch_size % (batch_gpu * num_gpus) == 0
        np.random.seed(random_seed * num_gpus + rank)
        torch.manual_seed(random_seed * num_gpus + rank)
        # allow_tf32 (training_loop_wo_flow_fullbody.py:243, 253-254; default False): the reference lets cuDNN / cuBLAS round
        # operands to TF32.  gfx950 has no TF32 matrix instructions; the counterpart here is the three-product split-bf16
        # arithmetic (2^-16 relative, against TF32's 2^-11) at half of the default mode's matrix work.
        if cfg.get('allow_tf32', False):
            from torch_utils.ops import conv2d_gradfix
            conv2d_gradfix.conv_math = 'bf16x3'

        self.G = dnnlib.util.construct_class_by_name(**cfg.G_kwargs).train().requires_grad_(False).to(device)
        self.D = dnnlib.util.construct_class_by_name(**cfg.D_kwargs).train().requires_grad_(False).to(device)
        self.G_ema = copy.deepcopy(self.G).eval()
        self.ema_kimg, self.ema_rampup = cfg.ema_kimg, cfg.ema_rampup

        # Replicas: rank 0's state everywhere, then a gradient exchange per optimised module.
        #   'flat'  - one FlatGradReducer for G and one for D (64 MiB buckets: G = 3 all-reduces, D = 2), gated by this
        #             class on the last accumulation round; the loss sees plain modules, its ddp_sync calls are no-ops.
        #   'torch' - the reference's arrangement (:316-324): one DistributedDataParallel wrapper per sub-module so that
        #             the loss can gate their all-reduces separately with ddp_sync; find_unused_parameters only where a
        #             parameter really takes no gradient (G.synthesis: b4.const; the parsing head when mask_weight = 0).
        G, D = self.G, self.D
        ddp = dict(G_mapping=G.mapping, G_synthesis=G.synthesis, G_const_encoding=G.const_encoding,
                   G_style_encoding=G.style_encoding, D=D)
        self.reducers = {}
        if num_gpus > 1 and ddp_mode == 'flat':
            with torch.no_grad():
                broadcast_module_states([G, D, self.G_ema])
            for name, module in [('G', G), ('D', D)]:
                if any(True for _ in module.parameters()):
                    self.reducers[name] = FlatGradReducer(module, num_gpus, bucket_mb=ddp_bucket_mb or 64)
        elif num_gpus > 1:
            for name, module in list(ddp.items()) + [(None, self.G_ema)]:
                if len(list(module.parameters())) != 0:
                    module.requires_grad_(True)
                    ids = [device] if device.type == 'cuda' else None
                    module = torch.nn.parallel.DistributedDataParallel(module, device_ids=ids, broadcast_buffers=False,
                                                                       find_unused_parameters=(name == 'G_synthesis'),
                                                                       bucket_cap_mb=ddp_bucket_mb or 25)
                    module.requires_grad_(False)
                if name is not None:
                    ddp[name] = module
        self.ddp_modules = ddp

        # ADA (training_loop_wo_flow_fullbody.py:301-310): the pipeline, its probability p, and the statistic that steers p
        self.augment_pipe = None
        self.ada_target, self.ada_interval, self.ada_kimg = cfg.get('ada_target'), cfg.get('ada_interval', 4), cfg.get('ada_kimg', 500)
        loss_kwargs = dict(cfg.loss_kwargs)
        if cfg.get('augment_kwargs') is not None and (cfg.get('augment_p', 0) > 0 or self.ada_target is not None):
            self.augment_pipe = dnnlib.util.construct_class_by_name(**cfg.augment_kwargs).train().requires_grad_(False).to(device)
            self.augment_pipe.p.copy_(torch.as_tensor(float(cfg.get('augment_p', 0))))
            if self.ada_target is not None:
                # sum and count of sign(D(real)) since the last adjustment, kept on the device (the reference's
                # training_stats.Collector(regex='Loss/signs/real') reads them back to the host every ada_interval)
                self._ada_acc = torch.zeros([2], device=device)
                user_report = loss_kwargs.get('report_fn')
                def report(name, value):
                    if name == 'Loss/signs/real':
                        v = value.detach().float()
                        self._ada_acc += torch.stack([v.sum(), torch.full([], float(v.numel()), device=v.device)])
                    if user_report is not None:
                        user_report(name, value)
                loss_kwargs['report_fn'] = report
        self.loss = dnnlib.util.construct_class_by_name(device=device, **ddp, augment_pipe=self.augment_pipe, **loss_kwargs)

        self.phases = []
        for name, module, opt_kwargs, reg_interval in [('G', G, cfg.G_opt_kwargs, cfg.G_reg_interval), ('D', D, cfg.D_opt_kwargs, cfg.D_reg_interval)]:
            if (device.type == 'cuda' and opt_kwargs.get('class_name') == 'torch.optim.Adam' and 'fused' not in opt_kwargs and 'foreach' not in opt_kwargs
                    and os.environ.get('PASTA_FUSED_ADAM', '1') != '0'):
                # the same update (torch.optim.Adam's formula) as ONE multi-tensor kernel per step instead of seven foreach passes
                opt_kwargs = dnnlib.EasyDict(opt_kwargs, fused=True)
            if reg_interval is None:
                opt = dnnlib.util.construct_class_by_name(params=module.parameters(), **opt_kwargs)
                self.phases += [dnnlib.EasyDict(name=name + 'both', module=module, opt=opt, interval=1)]
            else:   # lazy regularisation (:337-346)
                mb_ratio = reg_interval / (reg_interval + 1)
                opt_kwargs = dnnlib.EasyDict(opt_kwargs)
                opt_kwargs.lr = opt_kwargs.lr * mb_ratio
                opt_kwargs.betas = [beta ** mb_ratio for beta in opt_kwargs.betas]
                opt = dnnlib.util.construct_class_by_name(module.parameters(), **opt_kwargs)
                self.phases += [dnnlib.EasyDict(name=name + 'main', module=module, opt=opt, interval=1)]
                self.phases += [dnnlib.EasyDict(name=name + 'reg', module=module, opt=opt, interval=reg_interval)]
        self.batch_idx = 0
        self.cur_nimg = 0
        self._buf_versions = {}         # G buffer index -> version counter at its last copy into G_ema

    def run(self, data):
        """One iteration: every due phase accumulates gradients over the local rounds, then steps its optimiser;
        finally the EMA generator is updated. ``data`` is a ``SyntheticFullBodyBatch`` holding this rank's
        ``batch_size // num_gpus`` samples."""
        rounds = data.split(self.batch_gpu)
        z_dim = self.G.z_dim
        all_gen_z = torch.randn([len(self.phases), len(rounds) * self.batch_gpu, z_dim], device=self.device)
        for phase, phase_gen_z in zip(self.phases, all_gen_z):
            if self.batch_idx % phase.interval != 0:
                continue
            reducer = self.reducers.get(phase.name[0])
            phase.module.requires_grad_(True)
            if reducer is not None:
                reducer.begin()
            else:
                phase.opt.zero_grad(set_to_none=True)
            for round_idx, (r, gen_z) in enumerate(zip(rounds, phase_gen_z.split(self.batch_gpu))):
                sync = (round_idx == self.batch_size // (self.batch_gpu * self.num_gpus) - 1)
                if reducer is not None and sync:     # the exchange overlaps the backward pass(es) of the last round
                    reducer.arm(getattr(self.loss, 'backward_passes', lambda phase: None)(phase.name))
                self.loss.accumulate_gradients(phase=phase.name, gen_z=gen_z, sync=sync, gain=phase.interval, **r)
            phase.module.requires_grad_(False)
            if reducer is not None:
                reducer.finish()
                grads = reducer.flat_gradients()
            else:
                grads = [param.grad for param in phase.module.parameters() if param.grad is not None]
            if grads:       # nan_to_num(grad, nan=0, posinf=1e5, neginf=-1e5) (:513-515), one launch per 96 gradients
                misc.nan_to_num_(grads, nan=0, posinf=1e5, neginf=-1e5)
            phase.opt.step()

        ema_nimg = self.ema_kimg * 1000
        if self.ema_rampup is not None:
            ema_nimg = min(ema_nimg, self.cur_nimg * self.ema_rampup)
        ema_beta = 0.5 ** (self.batch_size / max(ema_nimg, 1e-8))
        with torch.no_grad():
            # p_ema <- p.lerp(p_ema, beta) (:522-529), as one multi-tensor launch: p_ema + (1 - beta) * (p - p_ema)
            torch._foreach_lerp_(list(self.G_ema.parameters()), list(self.G.parameters()), 1.0 - ema_beta)
            # b_ema.copy_(b) for every buffer (:528-529).  Only w_avg ever changes; a buffer whose version counter has not
            # moved since its last copy still equals its copy, so it is skipped (~75 tiny device copies per iteration).
            src, dst = [], []
            for i, (b_ema, b) in enumerate(zip(self.G_ema.buffers(), self.G.buffers())):
                if self._buf_versions.get(i) != b._version:
                    src.append(b); dst.append(b_ema)
                    self._buf_versions[i] = b._version
            if src:
                torch._foreach_copy_(dst, src)
        self.cur_nimg += self.batch_size

        # ADA adjustment (:536-539): p += sign(E[sign(D(real))] - target) * batch_size * ada_interval / (ada_kimg * 1000), p >= 0;
        # evaluated on the device, no read-back
        self.batch_idx += 1
        if self.augment_pipe is not None and self.ada_target is not None and self.batch_idx % self.ada_interval == 0:
            acc = self._ada_acc
            if self.num_gpus > 1:
                torch.distributed.all_reduce(acc)
            step = (self.batch_size * self.ada_interval) / (self.ada_kimg * 1000)
            mean = acc[0] / acc[1].clamp(min=1)
            adjust = torch.sign(mean - self.ada_target) * step * (acc[1] > 0)
            self.augment_pipe.p.copy_((self.augment_pipe.p + adjust).clamp(min=0))
            acc.zero_()

#----------------------------------------------------------------------------

def training_loop(num_gpus=1, rank=0, batch_size=16, batch_gpu=16, random_seed=0, total_iters=4, cfg=None, device=None, progress_fn=None):
    """Run ``total_iters`` iterations on synthetic data (the reference's loop runs until ``total_kimg``)."""
    device = device if device is not None else torch.device('cuda', rank)
    step = TrainingStep(device, cfg=cfg, num_gpus=num_gpus, rank=rank, batch_size=batch_size, batch_gpu=batch_gpu, random_seed=random_seed)
    data = SyntheticFullBodyBatch(batch_size // num_gpus, device, seed=rank)
    for it in range(total_iters):
        step.run(data)
        if progress_fn is not None:
            progress_fn(it + 1, total_iters)
    return step

#----------------------------------------------------------------------------
