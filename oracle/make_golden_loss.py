"""Training-step golden vectors (SURVEY.md 8c 'Training-step semantic'), build container only.

The reference's loss and loop modules cannot be imported here (they import cv2), so this harness drives the
REFERENCE'S OWN NETWORKS (training.networks, imported from /root/reference) with a hand-written restatement of
  * the loss terms of training/loss_wo_flow_fullbody.py:106-254 (non-saturating GAN terms on both generator outputs,
    L1 x l1_weight, parsing cross-entropy with class weights [1,2,2,3,3,3] x mask_weight, lazy R1), and
  * the hot loop of training/training_loop_wo_flow_fullbody.py:332-343 (phases, lazy-regularisation scaling of lr and
    betas) and :484-529 (zero_grad / requires_grad toggling / gain = interval / nan_to_num / Adam step / EMA),
and writes tests/golden/training_step.npz.  The overlay's StyleGAN2Loss and TrainingStep must reproduce it on the GPU
(tests/test_training_step_gpu.py).  Determinism: z_dim = 0 (style mixing is the identity) and every noise_strength is
set to 0, so the 'random' noise mode the loss runs the generator in has no effect on the values.

Called by ``oracle/make_golden.py --only loss``."""

import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, os.path.dirname(HERE))
from oracle import param_fill as PF  # noqa: E402

L1_WEIGHT, MASK_WEIGHT, R1_GAMMA = 40.0, 20.0, 10.0             # train.sh:3-10 (cfg fashion)
BATCH = 4                                                        # one mbstd group
DELTA_KEYS_G = ['synthesis.b4.conv1.weight', 'synthesis.b64.conv0.affine.weight', 'synthesis.b256.torgb.m_weight1',
                'synthesis.spade_b128_2.spade0.conv_gamma.weight', 'const_encoding.model.3.weight', 'style_encoding.fc.weight',
                'mapping.fc0.weight', 'synthesis.b128.merge_conv.weight', 'synthesis.b256.torgb.bias']
DELTA_KEYS_D = ['b256.fromrgb.weight', 'b64.conv1.weight', 'b8.skip.weight', 'b4.conv.weight', 'b4.fc.weight', 'mapping.fc3.bias', 'b4.out.weight']


def prepare(G, D):
    """Closed-form weights, noise switched off."""
    PF.fill_module(G); PF.fill_module(D)
    with torch.no_grad():
        for name, p in G.named_parameters():
            if name.endswith('noise_strength'):
                p.zero_()
    return G, D


def run_G(G, c, cat_feats_list, inp):
    cat_feats = {str(f.shape[2]): f for f in cat_feats_list}
    pose_feat = G.const_encoding(inp['pose'])
    ws = G.mapping(inp['gen_z'], c)
    return G.synthesis(ws, pose_feat, cat_feats, inp['denorm_upper_input'], inp['denorm_lower_input'],
                       inp['denorm_upper_mask'], inp['denorm_lower_mask'])


def accumulate(G, D, phase, inp, gain, log):
    """loss_wo_flow_fullbody.py:106-254 for vgg_weight = 0, pl_weight = 0."""
    softplus = torch.nn.functional.softplus
    ce = torch.nn.CrossEntropyLoss(ignore_index=255, weight=torch.tensor([1, 2, 2, 3, 3, 3], dtype=torch.float32))
    real_c, cat_feats = G.style_encoding(inp['style_input'], inp['retain'])
    gen_c = real_c
    if phase == 'Gmain':
        img, fin, parsing = run_G(G, gen_c, cat_feats, inp)
        lg, lf = D(img, gen_c), D(fin, gen_c)
        loss_Gmain, loss_Gmain_f = softplus(-lg).mean(), softplus(-lf).mean()
        l1 = torch.nn.L1Loss()(img, inp['real_img']) * L1_WEIGHT
        l1_f = torch.nn.L1Loss()(fin, inp['real_img']) * L1_WEIGHT
        mask = torch.mean(ce(parsing, inp['gt_parsing'].long()[:, 0, ...])) * MASK_WEIGHT
        loss_G = (loss_Gmain + loss_Gmain_f) / 2 + (l1 + l1_f) / 2 + mask
        log.update({'Loss/scores/fake': lg, 'Loss/scores/fake_finetune': lf, 'Loss/G/loss': loss_Gmain, 'Loss/G/loss_finetune': loss_Gmain_f,
                    'Loss/G/L1': l1, 'Loss/G/L1_finetune': l1_f, 'Loss/G/mask_loss': mask, 'total/G': loss_G})
        loss_G.mul(gain).backward()
    loss_Dgen = 0
    if phase == 'Dmain':
        img, fin, _ = run_G(G, gen_c, cat_feats, inp)
        lg, lf = D(img, gen_c), D(fin, gen_c)
        loss_Dgen, loss_Dgen_f = softplus(lg), softplus(lf)
        log.update({'Dmain/scores/fake': lg, 'Dmain/scores/fake_finetune': lf})
        ((loss_Dgen.mean() + loss_Dgen_f.mean()) / 2).mul(gain).backward()
    if phase in ('Dmain', 'Dreg'):
        do_r1 = phase == 'Dreg'
        real = inp['real_img'].detach().requires_grad_(do_r1)
        logits = D(real, real_c)
        log['Loss/scores/real' if not do_r1 else 'Dreg/scores/real'] = logits
        loss_Dreal, loss_Dr1 = 0, 0
        if phase == 'Dmain':
            loss_Dreal = softplus(-logits)
            log['Loss/D/loss'] = loss_Dgen + loss_Dreal
        if do_r1:
            # (the reference wraps this in conv2d_gradfix.no_weight_gradients(); on stock ops the weight gradients of
            # the first-order graph are simply not requested)
            g, = torch.autograd.grad(outputs=[logits.sum()], inputs=[real], create_graph=True, only_inputs=True)
            pen = g.square().sum([1, 2, 3])
            loss_Dr1 = pen * (R1_GAMMA / 2)
            log.update({'Loss/r1_penalty': pen, 'Loss/D/reg': loss_Dr1})
        (logits * 0 + loss_Dreal + loss_Dr1).mean().mul(gain).backward()


def gradnorms(module):
    return np.array([p.grad.norm().item() if p.grad is not None else -1.0 for _, p in sorted(module.named_parameters())])


def gen_loss(ref_root, import_reference_networks):
    rn = import_reference_networks(ref_root)
    torch.manual_seed(0)
    inp = PF.make_inputs(n=BATCH, seed=2)
    out = {}

    # ---- (a) one micro-batch per phase, gain 1: every reported scalar and the gradient norm of every parameter
    G, D = prepare(rn.GeneratorFull(**PF.G_KWARGS).train(), rn.Discriminator(**PF.D_KWARGS).train())
    for phase, module in [('Gmain', G), ('Dmain', D), ('Dreg', D)]:
        G.requires_grad_(False); D.requires_grad_(False)
        module.requires_grad_(True)
        module.zero_grad(set_to_none=True)
        log = {}
        accumulate(G, D, phase, inp, 1.0, log)
        for k, v in log.items():
            out[f'a.{phase}.{k}'] = v.detach().numpy().astype(np.float64)
        out[f'a.{phase}.gradnorms'] = gradnorms(module)
        print('phase', phase, 'done', flush=True)

    # ---- (b) two iterations of the loop on the same batch (batch_size = batch_gpu = 4, one GPU)
    G, D = prepare(rn.GeneratorFull(**PF.G_KWARGS).train().requires_grad_(False), rn.Discriminator(**PF.D_KWARGS).train().requires_grad_(False))
    G_ema = copy.deepcopy(G).eval()
    init_G = {k: v.detach().clone() for k, v in G.named_parameters()}
    init_D = {k: v.detach().clone() for k, v in D.named_parameters()}
    phases = []
    for name, module, interval in [('G', G, 4), ('D', D, 16)]:
        mb = interval / (interval + 1)
        opt = torch.optim.Adam(module.parameters(), lr=0.002 * mb, betas=[0 ** mb, 0.99 ** mb], eps=1e-8)
        phases += [dict(name=name + 'main', module=module, opt=opt, interval=1), dict(name=name + 'reg', module=module, opt=opt, interval=interval)]
    cur_nimg = 0
    for batch_idx in range(2):
        for ph in phases:
            if batch_idx % ph['interval'] != 0:
                continue
            ph['opt'].zero_grad(set_to_none=True)
            ph['module'].requires_grad_(True)
            if ph['name'] != 'Greg':                       # pl_weight = 0: the phase only runs the style encoder forward
                accumulate(G, D, ph['name'], inp, float(ph['interval']), {})
            else:
                G.style_encoding(inp['style_input'], inp['retain'])
            ph['module'].requires_grad_(False)
            for p in ph['module'].parameters():
                if p.grad is not None:
                    torch.nan_to_num(p.grad, nan=0, posinf=1e5, neginf=-1e5, out=p.grad)
            ph['opt'].step()
        ema_beta = 0.5 ** (BATCH / max(10 * 1000, 1e-8))
        with torch.no_grad():
            for p_ema, p in zip(G_ema.parameters(), G.parameters()):
                p_ema.copy_(p.lerp(p_ema, ema_beta))
            for b_ema, b in zip(G_ema.buffers(), G.buffers()):
                b_ema.copy_(b)
        cur_nimg += BATCH
        print('iteration', batch_idx, 'done', flush=True)
    gd, dd, ed = dict(G.named_parameters()), dict(D.named_parameters()), dict(G_ema.named_parameters())
    for k in DELTA_KEYS_G:
        out['b.G.delta.' + k] = PF.summarize(gd[k].detach() - init_G[k])['sample']
        out['b.G_ema.delta.' + k] = PF.summarize(ed[k].detach() - init_G[k])['sample']
    for k in DELTA_KEYS_D:
        out['b.D.delta.' + k] = PF.summarize(dd[k].detach() - init_D[k])['sample']
    out['b.G.w_avg'] = G.mapping.w_avg.detach().numpy()
    out['b.G_ema.w_avg'] = G_ema.mapping.w_avg.detach().numpy()
    np.savez_compressed(os.path.join(GOLDEN, 'training_step.npz'), **out)
    print('training-step fixtures written:', len(out), 'arrays')
