"""Snapshot format: persistent classes pickle in the reference's layout, pickled classes re-bind by
name to this package's classes without executing the embedded source, import hooks see and may edit the metadata, and
``legacy.load_network_pkl`` returns the snapshot dictionary test.py expects."""

import io
import pickle

import pytest
import torch

from oracle import param_fill as PF


def _tiny_generator(cls='GeneratorV18'):
    from training import networks
    kw = dict(PF.G_KWARGS)
    kw['synthesis_kwargs'] = dict(channel_base=512, channel_max=32, conv_clamp=256)
    return PF.fill_module(getattr(networks, cls)(**kw)).eval().requires_grad_(False)


def test_round_trip_keeps_class_arguments_and_weights():
    from torch_utils import persistence
    G = _tiny_generator()
    blob = pickle.dumps(G)
    G2 = pickle.loads(blob)
    assert type(G2) is type(G) and persistence.is_persistent(G2)
    assert G2.init_kwargs == G.init_kwargs and G2.init_args == G.init_args
    sd, sd2 = G.state_dict(), G2.state_dict()
    assert list(sd) == list(sd2) and all(torch.equal(sd[k], sd2[k]) for k in sd)
    assert not G2.training
    # nested persistent modules are persistent objects of their own
    assert persistence.is_persistent(G2.synthesis) and persistence.is_persistent(G2.mapping.fc0)


def test_pickle_layout_is_the_reference_format():
    from torch_utils import persistence
    G = _tiny_generator()
    fn, (meta,), state = G.__reduce__()
    assert fn is persistence._reconstruct_persistent_obj and state is None
    assert set(meta) == {'type', 'version', 'module_src', 'class_name', 'state'}
    assert meta['type'] == 'class' and meta['version'] == 6 and meta['class_name'] == 'GeneratorV18'
    assert 'class GeneratorV18' in meta['module_src']                     # source text of the defining module
    assert {'_parameters', '_buffers', '_modules', '_init_args', '_init_kwargs'} <= set(meta['state'])


def test_foreign_snapshot_binds_to_local_classes_without_running_its_source():
    """A snapshot written elsewhere carries that writer's module text; loading must not execute it (the reference's
    networks.py raises at import on ROCm) but bind every class name to the local implementation."""
    from torch_utils import persistence
    G = _tiny_generator()
    seen = []

    def poison(meta):       # what a reference-written pickle looks like to this loader: unknown, unimportable source
        seen.append(meta.class_name)
        meta.module_src = 'raise RuntimeError("the embedded source must not be executed")\n'
        return meta

    persistence._import_hooks.append(poison)
    try:
        G2 = pickle.loads(pickle.dumps(G))
    finally:
        persistence._import_hooks.remove(poison)
    assert type(G2) is type(G)
    assert 'GeneratorV18' in seen and 'SynthesisLayer' in seen and 'FullyConnectedLayer' in seen
    sd, sd2 = G.state_dict(), G2.state_dict()
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)


def test_import_hook_can_edit_the_state():
    from torch_utils import persistence
    from training import networks
    layer = PF.fill_module(networks.FullyConnectedLayer(8, 4))

    @persistence.import_hook
    def zero_bias(meta):
        if meta.class_name == 'FullyConnectedLayer':
            meta.state['_parameters']['bias'] = torch.nn.Parameter(torch.zeros_like(meta.state['_parameters']['bias']))
        return meta
    try:
        layer2 = pickle.loads(pickle.dumps(layer))
    finally:
        persistence._import_hooks.remove(zero_bias)
    assert torch.equal(layer2.weight, layer.weight) and float(layer2.bias.abs().sum()) == 0.0 and float(layer.bias.abs().sum()) > 0


def test_unknown_class_name_falls_back_to_the_embedded_source():
    from torch_utils import persistence
    src = ("import torch\nfrom torch_utils import persistence\n"
           "@persistence.persistent_class\nclass OnlyInTheSnapshot(torch.nn.Module):\n"
           "    def __init__(self, n):\n        super().__init__()\n        self.w = torch.nn.Parameter(torch.ones([n]))\n")
    meta = dict(type='class', version=6, module_src=src, class_name='OnlyInTheSnapshot',
                state=dict(_parameters={'w': torch.nn.Parameter(torch.full([3], 2.0))}, _buffers={}, _modules={}, _init_args=(3,), _init_kwargs={},
                           training=True, _backward_hooks={}, _forward_hooks={}, _forward_pre_hooks={}))
    obj = persistence._reconstruct_persistent_obj(meta)
    assert type(obj).__name__ == 'OnlyInTheSnapshot' and float(obj.w.sum()) == 6.0
    assert 'OnlyInTheSnapshot' not in persistence._local_classes


def test_load_network_pkl():
    import legacy
    G = _tiny_generator()
    from training import networks
    D = PF.fill_module(networks.Discriminator(c_dim=512, img_resolution=256, img_channels=3, channel_base=512, channel_max=32, conv_clamp=256))
    buf = io.BytesIO()
    pickle.dump(dict(G=G, D=D, G_ema=G), buf)
    buf.seek(0)
    data = legacy.load_network_pkl(buf)
    assert set(data) >= {'G', 'D', 'G_ema', 'training_set_kwargs', 'augment_pipe'}
    assert data['training_set_kwargs'] is None and data['augment_pipe'] is None
    assert type(data['G_ema']) is type(G) and type(data['D']) is type(D)
    with pytest.raises(ValueError):
        legacy.load_network_pkl(io.BytesIO(pickle.dumps([1, 2, 3])))
    # force_fp16 (reference legacy.py:45-59): every network is rebuilt with num_fp16_res = 4, conv_clamp = 256 and the same weights
    buf.seek(0)
    half = legacy.load_network_pkl(buf, force_fp16=True)
    assert half['D'].init_kwargs['num_fp16_res'] == 4 and half['D'].init_kwargs['conv_clamp'] == 256 and half['D'].b256.use_fp16
    assert half['G_ema'].init_kwargs['synthesis_kwargs']['num_fp16_res'] == 4 and G.init_kwargs['synthesis_kwargs'].get('num_fp16_res') is None
    assert type(half['G']) is type(G) and not half['G'].training
    sd, sd2 = D.state_dict(), half['D'].state_dict()
    assert list(sd) == list(sd2) and all(torch.equal(sd[k], sd2[k]) for k in sd)


@pytest.mark.gpu
def test_loaded_generator_runs_on_the_hip_path():
    """test.py's flow: load the snapshot, move G_ema to the GPU, run the inference call sequence."""
    import legacy
    from training import networks
    G = PF.fill_module(networks.GeneratorV18(**PF.G_KWARGS)).eval().requires_grad_(False)      # a configuration that can run
    buf = io.BytesIO()
    D = networks.Discriminator(c_dim=512, img_resolution=256, img_channels=3, channel_base=512, channel_max=32)
    pickle.dump(dict(G=G, D=D, G_ema=G), buf)
    buf.seek(0)
    G2 = legacy.load_network_pkl(buf)['G_ema'].cuda()
    inp = {k: v.cuda() for k, v in PF.make_inputs(n=2, seed=0).items()}
    c60 = PF.make_inputs(n=2, seed=5)['style_input'].repeat(1, 2, 1, 1)[:, :60].cuda()
    G = G.cuda()
    with torch.no_grad():
        a = G(inp['gen_z'], c60, inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
              inp['denorm_upper_mask'], inp['denorm_lower_mask'], noise_mode='const')
        b = G2(inp['gen_z'], c60, inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
               inp['denorm_upper_mask'], inp['denorm_lower_mask'], noise_mode='const')
    for x, y in zip(a, b):
        assert torch.equal(x, y)


# ---- a snapshot written by the reference's own classes (oracle/make_golden_snapshot.py) --------------------------------

def _reference_snapshot():
    import legacy
    from conftest import GOLDEN
    import os
    with open(os.path.join(GOLDEN, 'reference_snapshot.pkl'), 'rb') as f:
        return legacy.load_network_pkl(f)


def test_reference_written_snapshot_binds_to_local_classes():
    """``legacy.load_network_pkl`` on a pickle produced by the reference's persistence + networks modules (reference
    persistence.py:99-126, training_loop_wo_flow_fullbody.py:588-602): every pickled class name resolves to this package's
    class, constructor arguments survive, and the tensors are the ones the reference stored."""
    from oracle import make_golden_snapshot as MS
    from torch_utils import persistence
    from training import networks
    data = _reference_snapshot()
    assert data['training_set_kwargs']['class_name'] == 'training.dataset.UvitonDatasetFull' and data['augment_pipe'] is None
    D, block, spade = data['D'], data['G'], data['G_ema']
    assert type(D) is networks.Discriminator and type(block) is networks.SynthesisBlockFull and type(spade) is networks.Spade_ResBlockV2
    assert type(D.b64.conv1) is networks.Conv2dLayer and type(D.mapping.fc3) is networks.FullyConnectedLayer
    assert type(block.torgb) is networks.ToRGBLayerFull and type(block.conv0) is networks.SynthesisLayer
    assert type(spade.spade0) is networks.Spade_Norm_Block and type(spade.spade0.conv_mlp) is networks.Spade_Conv2dLayer
    assert all(persistence.is_persistent(m) for m in (D, D.b4, D.b4.mbstd, block.merge_conv, spade.skip))
    assert dict(D.init_kwargs) == MS.D_KWARGS and dict(block.init_kwargs) == MS.BLOCK_ARGS and dict(spade.init_kwargs) == MS.SPADE_ARGS
    assert not D.training and not any(p.requires_grad for p in D.parameters())
    # the weights are the closed form the generator script filled in: a locally constructed twin, filled the same way, is equal
    for loaded, twin in [(D, networks.Discriminator(**MS.D_KWARGS)), (block, networks.SynthesisBlockFull(**MS.BLOCK_ARGS)),
                         (spade, networks.Spade_ResBlockV2(**MS.SPADE_ARGS))]:
        want = PF.fill_module(twin).state_dict()
        got = loaded.state_dict()
        assert list(got) == list(want)
        assert all(torch.equal(got[k], want[k]) for k in want)
    # derived attributes come from the local constructor, not from the pickled __dict__
    assert block.torgb._heads == (('1', 6, 'linear'),) and D.b256.num_layers == 4


@pytest.mark.gpu
def test_reference_written_snapshot_reproduces_the_reference_outputs():
    from conftest import load_golden, rel_err
    from oracle import make_golden_snapshot as MS
    g = load_golden('reference_snapshot_outputs.npz')
    data = _reference_snapshot()
    inp = {k: v.cuda() for k, v in MS.snapshot_inputs().items()}
    D, block, spade = data['D'].cuda(), data['G'].cuda(), data['G_ema'].cuda()
    with torch.no_grad():
        assert rel_err(D(inp['d_img'], inp['d_c']), g['D.logits']) < 1e-4
        x, img, parsing = block(inp['b_x'], inp['b_img'], inp['b_ws'], None, {'32': inp['b_cat']}, noise_mode='const', fused_modconv=False)
        assert rel_err(x, g['G.x']) < 1e-4 and rel_err(img, g['G.img']) < 1e-4 and rel_err(parsing, g['G.parsing']) < 1e-4
        assert rel_err(spade(inp['s_x'], inp['s_feat']), g['G_ema.y']) < 1e-4
