"""Training-step semantics on the GPU against tests/golden/training_step.npz (oracle/make_golden_loss.py: the
reference's own networks driven by a restatement of its loss and hot loop).

(a) StyleGAN2Loss.accumulate_gradients: every reported scalar and the gradient norm of every parameter, per phase.
(b) TrainingStep.run x 2: parameter updates of G, D and G_ema (phase schedule, lazy-regularisation scaling of lr and
    betas, gain = interval, nan_to_num, Adam, EMA)."""

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import param_fill as PF
from oracle.make_golden_loss import BATCH, DELTA_KEYS_D, DELTA_KEYS_G, prepare

pytestmark = pytest.mark.gpu

TOL_SCALAR = 2e-4       # losses are means over <= 4 x 256 x 256 values of fp32 network outputs
TOL_GRAD = 1e-3         # BASELINE.json's bar for gradients


def _batch():
    inp = PF.make_inputs(n=BATCH, seed=2)
    return {k: v.cuda() for k, v in inp.items()}


def _kwargs(inp):
    keys = ['real_img', 'gen_z', 'style_input', 'retain', 'pose', 'denorm_upper_input', 'denorm_lower_input', 'denorm_upper_mask',
            'denorm_lower_mask', 'gt_parsing']
    return {k: inp[k] for k in keys}


@pytest.mark.parametrize('phase', ['Gmain', 'Dmain', 'Dreg'])
def test_loss_terms_and_gradients(phase):
    from training import networks
    from training.loss_wo_flow_fullbody import StyleGAN2Loss
    g = load_golden('training_step.npz')
    G, D = prepare(networks.GeneratorFull(**PF.G_KWARGS).train(), networks.Discriminator(**PF.D_KWARGS).train())
    G.cuda().requires_grad_(False); D.cuda().requires_grad_(False)
    log = {}
    loss = StyleGAN2Loss(torch.device('cuda'), G.mapping, G.synthesis, G.const_encoding, G.style_encoding, D, r1_gamma=10, l1_weight=40,
                         vgg_weight=0, contextual_weight=0, pl_weight=0, mask_weight=20, report_fn=lambda name, value: log.__setitem__(name, value))
    module = G if phase == 'Gmain' else D
    module.requires_grad_(True)
    loss.accumulate_gradients(phase=phase, sync=True, gain=1.0, **_kwargs(_batch()))
    rename = {'Dmain': {}, 'Dreg': {'Loss/scores/real': 'Dreg/scores/real'}, 'Gmain': {}}[phase]
    checked = 0
    for name, value in log.items():
        key = f'a.{phase}.{rename.get(name, name)}'
        if key in g:
            v = torch.as_tensor(value).detach().double().cpu().numpy()
            assert np.allclose(v, g[key], rtol=TOL_SCALAR, atol=TOL_SCALAR * float(np.abs(g[key]).max())), (key, v, g[key])
            checked += 1
    assert checked >= {'Gmain': 7, 'Dmain': 2, 'Dreg': 3}[phase], sorted(log)
    norms = np.array([p.grad.norm().item() if p.grad is not None else -1.0 for _, p in sorted(module.named_parameters())])
    ref = g[f'a.{phase}.gradnorms']
    assert norms.shape == ref.shape
    assert ((norms < 0) == (ref < 0)).all()                     # the same parameters receive a gradient
    names = [n for n, _ in sorted(module.named_parameters())]
    # not comparable: d/d noise_strength = <dy, noise> with freshly drawn noise (the loss runs G in 'random' noise mode);
    # the style encoder's pyramid weights, whose gradients are instance-norm cancellation residue
    skip = np.array([n.endswith('noise_strength') or n.startswith('style_encoding.model.') for n in names])
    big = (ref > 1e-6 * ref.max()) & ~skip
    assert big.sum() >= 0.5 * ((ref >= 0) & ~skip).sum()
    worst = int(np.argmax(np.abs(norms - ref) * big))
    assert np.abs(norms[big] - ref[big]).max() / ref.max() < TOL_GRAD, (names[worst], norms[worst], ref[worst])
    assert np.abs(norms[big] / ref[big] - 1).max() < 5 * TOL_GRAD, names[int(np.argmax(np.abs(norms / np.maximum(ref, 1e-30) - 1) * big))]
    assert np.median(np.abs(norms[big] / ref[big] - 1)) < 1e-4


def test_two_iterations_of_the_loop(arith):
    from training.training_loop_wo_flow_fullbody import TrainingStep, fashion_config

    class Batch:                                   # TrainingStep only needs split()
        def __init__(self, inp):
            self.inp = inp
        def split(self, n):
            keys = [k for k in self.inp if k != 'gen_z']
            return [{k: self.inp[k][i:i + n] for k in keys} for i in range(0, BATCH, n)]

    g = load_golden('training_step.npz')
    step = TrainingStep(torch.device('cuda'), cfg=fashion_config(channel_base=2048), num_gpus=1, rank=0, batch_size=BATCH, batch_gpu=BATCH)
    prepare(step.G, step.D)
    step.G_ema.load_state_dict(step.G.state_dict())
    init_G = {k: v.detach().clone() for k, v in step.G.named_parameters()}
    init_D = {k: v.detach().clone() for k, v in step.D.named_parameters()}
    data = Batch(_batch())
    step.run(data)
    step.run(data)
    assert step.batch_idx == 2 and step.cur_nimg == 2 * BATCH

    def check(tag, params, init, keys):
        for k in keys:
            ours = PF.summarize(params[k].detach() - init[k])['sample'].astype(np.float64)
            ref = g[f'{tag}.delta.{k}'].astype(np.float64)
            # Adam's first steps move every element by about lr whatever the gradient's size, so elements whose gradient
            # is rounding noise may step the other way: compare direction and size of the update, not element values.
            if np.linalg.norm(ref) == 0:             # a parameter the step leaves alone (zero gradient): same here
                assert np.linalg.norm(ours) == 0, (tag, k)
                continue
            cos = float((ours * ref).sum() / (np.linalg.norm(ours) * np.linalg.norm(ref)))
            size = abs(np.abs(ours).mean() / np.abs(ref).mean() - 1)
            print(f'{tag} {k}: cosine {cos:.4f}, size {size:.4f}')
            assert cos > 0.98, (tag, k, cos)
            # measured worst key (style_encoding.fc.weight, whose gradient is instance-norm cancellation residue, so many of its
            # elements step by +-lr on rounding noise) 0.6 % fp32 MFMA, 1.9 % split-bf16, 2.7 % fp16 x 3 (profiles/r3_two_iterations_by_arithmetic.txt):
            # split-bf16 stays at the 2 % it met before the default changed, the three-product default is given 3 %
            assert size < (0.03 if arith == 'f16x3' else 0.02), (tag, k, size, arith)

    check('b.G', dict(step.G.named_parameters()), init_G, DELTA_KEYS_G)
    check('b.D', dict(step.D.named_parameters()), init_D, DELTA_KEYS_D)
    check('b.G_ema', dict(step.G_ema.named_parameters()), init_G, DELTA_KEYS_G)
    # w_avg: four EMA updates per iteration pair, through mapping weights that already took Adam steps (see above)
    assert rel_err(step.G.mapping.w_avg, g['b.G.w_avg']) < 2e-3
    assert rel_err(step.G_ema.mapping.w_avg, g['b.G_ema.w_avg']) < 2e-3
    # EMA is a fixed fraction of the live update: p_ema - p0 = (1 - beta)(1 + beta) / 2 ... of two different steps; at least
    # it must be much smaller than the live update and non-zero
    k = DELTA_KEYS_G[0]
    d_live = (dict(step.G.named_parameters())[k] - init_G[k]).abs().mean().item()
    d_ema = (dict(step.G_ema.named_parameters())[k] - init_G[k]).abs().mean().item()
    assert 0 < d_ema < 0.01 * d_live


def _forced_slopes(tape):
    """Context: torch_utils.ops.bias_act.slope_tape = tape (the test instrument that removes leaky-ReLU slope flips from a comparison)."""
    import contextlib
    from torch_utils.ops import bias_act

    @contextlib.contextmanager
    def ctx():
        old, bias_act.slope_tape = bias_act.slope_tape, tape
        try:
            yield tape
        finally:
            bias_act.slope_tape = old
    return ctx()


def test_merged_discriminator_pass_equals_separate_passes(arith):
    """run_D_multi: one discriminator pass over several image batches, stacked so that every minibatch-std group stays
    inside its own batch, returns the logits (and, through them, the gradients) of the separate passes.

    The two ways take different launch plans (K slices at 8 images, none at 24), each fp32-accurate to ~1e-6 per convolution; through
    14 layers the image gradients differ by 3e-6 .. 1e-5 of their maximum -- unless a leaky-ReLU pre-activation within rounding of zero
    takes the other slope in one of the two passes, which moves ONE sample's gradient by as much as that unit's share of it (measured up
    to 5 %; profiles/r4_merged_d_flips.txt) and WHICH samples flip moves with every change of a rounding anywhere in the discriminator:
    rounds 3 and 4 re-toleranced this test twice for it ("<= 4 of 72 samples beyond 3e-4").  Round 5: the separate
    passes are evaluated with the slopes the merged pass took (bias_act.SlopeTape: sign masks recorded in call order, the outputs that
    differ moved across zero before they are saved), so both sides differentiate the same piecewise-linear function, and EVERY sample of
    every seed is held to 3e-4 under BOTH arithmetics; what was moved must be rounding-sized."""
    from training import networks
    from training.loss_wo_flow_fullbody import StyleGAN2Loss
    from torch_utils.ops import bias_act
    G, D = prepare(networks.GeneratorFull(**PF.G_KWARGS).train(), networks.Discriminator(**PF.D_KWARGS).train())
    D.cuda()
    loss = StyleGAN2Loss(torch.device('cuda'), G.mapping, G.synthesis, G.const_encoding, G.style_encoding, D, vgg_weight=0, contextual_weight=0)
    assert loss._mbstd_groups(16) == 4 and loss._mbstd_groups(8) == 2 and loss._mbstd_groups(6) is None
    n, k = 8, 3
    B = loss._mbstd_groups(n); Gr = n // B
    all_samples, moved = [], 0
    for seed in (9, 1, 2):
        g = torch.Generator().manual_seed(seed)
        imgs = [(torch.rand([n, 3, 256, 256], generator=g) * 2 - 1).cuda().requires_grad_(True) for _ in range(k)]
        cs = [torch.randn([n, 512], generator=g).cuda() for _ in range(k)]
        with _forced_slopes(bias_act.SlopeTape()) as rec:
            mer = loss.run_D_multi(imgs, cs, sync=True)
        assert len(rec.masks) >= 20                         # every lrelu of the mapping network, the blocks and the epilogue
        sep = []
        for j in range(k):                                  # batch j's samples sit at [:, j] of the merged pass's [G, k, B] arrangement
            pick = lambda m, j=j: m.reshape(Gr, k, B, *m.shape[1:])[:, j].reshape(n, *m.shape[1:])
            with _forced_slopes(bias_act.SlopeTape(replay=rec.masks, select=pick)) as rep:
                sep.append(loss.run_D(imgs[j], cs[j], sync=True))
            assert rep.pos == len(rec.masks)                # the same activations in the same order
            assert rep.worst < 1e-5, (arith, seed, j, rep.worst)      # only pre-activations within rounding of zero were moved
            moved += rep.moved
        for a, b in zip(sep, mer):
            assert a.shape == b.shape == (n, 1)
            assert rel_err(b, a) < 1e-5
        w = [torch.randn([n, 1], generator=g).cuda() for _ in range(k)]
        g_sep = torch.autograd.grad(sum((a * x).sum() for a, x in zip(sep, w)), imgs)
        g_mer = torch.autograd.grad(sum((a * x).sum() for a, x in zip(mer, w)), imgs)
        for a, b in zip(g_sep, g_mer):
            pm = a.abs().amax(dim=[1, 2, 3]).clamp_min(1e-300)
            per_sample = ((b - a).abs().amax(dim=[1, 2, 3]) / pm).cpu()
            assert float(per_sample.max()) < 3e-4, (arith, seed, per_sample)
            all_samples += per_sample.tolist()
    print(f'{arith}: {moved} activations moved across zero over {len(all_samples)} samples, worst sample {max(all_samples):.2e}')
    assert sorted(all_samples)[len(all_samples) // 2] < 1e-5, (arith, sorted(all_samples)[len(all_samples) // 2])


def test_joined_gradients_and_fused_residual_sums_equal_the_plain_graph_in_the_discriminator():
    """the deterministic check of the joined-gradient path itself -- the discriminator with and without
    ``networks._GRAD_JOIN`` / ``_SKIP_ADD_FUSED`` (the residual sum in the skip convolution's epilogue, the skip branch's gradient as the
    residual of conv0's input-gradient launch) on the same batch, slopes forced equal: the image gradients agree on every sample."""
    from training import networks
    from torch_utils.ops import bias_act
    _, D = prepare(networks.GeneratorFull(**PF.G_KWARGS).train(), networks.Discriminator(**PF.D_KWARGS).train())
    D.cuda()
    g = torch.Generator().manual_seed(4)
    img = (torch.rand([8, 3, 256, 256], generator=g) * 2 - 1).cuda().requires_grad_(True)
    c = torch.randn([8, 512], generator=g).cuda()
    w = torch.randn([8, 1], generator=g).cuda()

    def run(flags, tape):
        old = networks._GRAD_JOIN, networks._SKIP_ADD_FUSED
        networks._GRAD_JOIN, networks._SKIP_ADD_FUSED = flags
        try:
            with _forced_slopes(tape):
                logits = D(img, c)
            gx, = torch.autograd.grad((logits * w).sum(), [img])
            return logits, gx
        finally:
            networks._GRAD_JOIN, networks._SKIP_ADD_FUSED = old
    rec = bias_act.SlopeTape()
    l1, g1 = run((True, True), rec)
    rep = bias_act.SlopeTape(replay=rec.masks)
    l0, g0 = run((False, False), rep)
    assert rep.pos == len(rec.masks) and rep.worst < 1e-5
    assert rel_err(l1, l0) < 1e-5
    pm = g0.abs().amax(dim=[1, 2, 3])
    per_sample = ((g1 - g0).abs().amax(dim=[1, 2, 3]) / pm).cpu()
    assert float(per_sample.max()) < 2e-5, per_sample


def test_ada_controller_moves_p_like_the_reference():
    """Training with --aug ada (train_wo_flow_fullbody.py:257-313): the pipeline is applied to every discriminator input,
    and every ada_interval iterations p moves by sign(mean(sign(D(real))) - target) * batch * interval / (ada_kimg * 1000),
    never below 0 (training_loop_wo_flow_fullbody.py:536-539) -- computed on the device here, recomputed on the host from
    the reported logits below."""
    from training.training_loop_wo_flow_fullbody import TrainingStep, fashion_config, augment_options

    class Batch:
        def __init__(self, inp):
            self.inp = inp
        def split(self, n):
            keys = [k for k in self.inp if k != 'gen_z']
            return [{k: self.inp[k][i:i + n] for k in keys} for i in range(0, BATCH, n)]

    signs = []
    cfg = fashion_config(channel_base=2048)
    cfg.update(augment_options(aug='ada', augpipe='bgc', p=0.3, target=0.6))
    cfg.ada_interval, cfg.ada_kimg = 2, 0.5                    # step = 4 * 2 / 500 = 0.016 per adjustment
    cfg.loss_kwargs.report_fn = lambda name, value: signs.append(value.detach()) if name == 'Loss/signs/real' else None
    step = TrainingStep(torch.device('cuda'), cfg=cfg, num_gpus=1, rank=0, batch_size=BATCH, batch_gpu=BATCH)
    assert type(step.augment_pipe).__name__ == 'AugmentPipe' and step.loss.augment_pipe is step.augment_pipe
    assert abs(float(step.augment_pipe.p) - 0.3) < 1e-7
    prepare(step.G, step.D)
    data = Batch(_batch())
    p_host, mark = 0.3, 0
    for it in range(5):
        step.run(data)
        if (it + 1) % 2 == 0:     # batch_idx is incremented first (training_loop...:532-538): adjustments after iterations 1 and 3,
                                  # each using everything reported since the previous one
            pending = torch.cat([s.flatten() for s in signs[mark:]])
            mark = len(signs)
            p_host = max(p_host + float(np.sign(pending.mean().item() - 0.6)) * (BATCH * 2) / (0.5 * 1000), 0.0)
        assert abs(float(step.augment_pipe.p) - p_host) < 1e-6, (it, float(step.augment_pipe.p), p_host)
    assert p_host != 0.3
    for prm in list(step.G.parameters()) + list(step.D.parameters()):
        assert bool(torch.isfinite(prm).all())


def test_allow_tf32_selects_the_three_product_arithmetic():
    """allow_tf32 (training_loop_wo_flow_fullbody.py:243, 253-254): the counterpart here is PASTA_MATH_BF16X3; a step in that
    mode lands within 1e-3 of the default step's parameters after one iteration (the north-star parity bar)."""
    from torch_utils.ops import conv2d_gradfix
    from training.training_loop_wo_flow_fullbody import TrainingStep, fashion_config

    class Batch:
        def __init__(self, inp):
            self.inp = inp
        def split(self, n):
            keys = [k for k in self.inp if k != 'gen_z']
            return [{k: self.inp[k][i:i + n] for k in keys} for i in range(0, BATCH, n)]

    old = conv2d_gradfix.conv_math
    try:
        logs = {}
        for allow in (False, True):
            conv2d_gradfix.conv_math = 'default'
            cfg = fashion_config(channel_base=2048)
            cfg.allow_tf32 = allow
            log = {}
            cfg.loss_kwargs.report_fn = lambda name, value, log=log: log.__setitem__(name, torch.as_tensor(value).detach().float().mean().item())
            step = TrainingStep(torch.device('cuda'), cfg=cfg, num_gpus=1, rank=0, batch_size=BATCH, batch_gpu=BATCH)
            assert conv2d_gradfix.conv_math == ('bf16x3' if allow else 'default')
            prepare(step.G, step.D)
            step.run(Batch(_batch()))
            logs[allow] = log
        for k, v in logs[False].items():
            assert abs(logs[True][k] - v) <= 1e-3 * max(abs(v), 1e-3), (k, v, logs[True][k])
        assert any(logs[True][k] != v for k, v in logs[False].items())        # the arithmetic really differed
    finally:
        conv2d_gradfix.conv_math = old
