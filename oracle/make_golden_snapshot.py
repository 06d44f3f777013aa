"""A network snapshot WRITTEN BY THE REFERENCE'S OWN CLASSES (build container only; ``oracle/make_golden.py --only snapshot``).
TEST INFRASTRUCTURE.

The reference saves ``dict(G=..., D=..., G_ema=..., augment_pipe=..., training_set_kwargs=...)`` with ``pickle.dump``
(training_loop_wo_flow_fullbody.py:588-602); every persistent object reduces to
``torch_utils.persistence._reconstruct_persistent_obj(meta)`` with ``meta = dict(type, version, module_src, class_name,
state)`` (persistence.py:99-126).  This script builds small instances of the reference's classes, fills them with the
closed-form weights of ``oracle/param_fill.py``, runs them on seeded inputs on the CPU, and writes

* ``tests/golden/reference_snapshot.pkl`` -- the pickle, with ``module_src`` BLANKED: that field is the full text of the
  reference's ``training/networks.py``, which must not enter this repository.  The loader under test never reads it when
  a local class of the pickled name exists (torch_utils/persistence.py; tests/test_persistence.py poisons it to prove that);
* ``tests/golden/reference_snapshot_outputs.npz`` -- the reference's outputs for those inputs.

The full generator cannot be a fixture (its encoders have fixed widths: 60 MB); the snapshot therefore holds the
discriminator at narrow widths under ``D`` and, under ``G`` / ``G_ema``, two generator sub-networks that together nest every
persistent layer class the generator is made of.
"""

import os
import pickle
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, os.path.dirname(HERE))
from oracle import param_fill as PF  # noqa: E402

D_KWARGS = dict(c_dim=512, img_resolution=256, img_channels=3, channel_base=256, channel_max=16, conv_clamp=256)
BLOCK_ARGS = dict(in_channels=16, out_channels=8, w_dim=32, resolution=32, img_channels=3, is_last=True, is_style=True, conv_clamp=256)
SPADE_ARGS = dict(in_channels=4, out_channels=4)


def snapshot_inputs():
    g = torch.Generator().manual_seed(77)
    r = lambda *shape: torch.randn(shape, generator=g)
    return dict(d_img=PF.make_inputs(n=4, seed=1)['real_img'], d_c=torch.tanh(r(4, 512)),
                b_x=r(2, 16, 16, 16), b_img=r(2, 3, 16, 16), b_ws=r(2, 3, 32), b_cat=r(2, 64, 32, 32),
                s_x=r(2, 4, 32, 32) * 0.5, s_feat=r(2, 256, 32, 32) * 0.5)


def gen_snapshot(ref_root, import_reference_networks):
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    os.chdir(ref_root)
    import torch_utils.persistence as ref_persistence
    ref_persistence._module_to_src = lambda module: ''          # see the module docstring: no reference text in the fixture
    rn = import_reference_networks(ref_root)
    torch.manual_seed(0)
    D = PF.fill_module(rn.Discriminator(**D_KWARGS)).eval().requires_grad_(False)
    block = PF.fill_module(rn.SynthesisBlockFull(**BLOCK_ARGS)).eval().requires_grad_(False)
    spade = PF.fill_module(rn.Spade_ResBlockV2(**SPADE_ARGS)).eval().requires_grad_(False)
    inp = snapshot_inputs()
    out = {}
    with torch.no_grad():
        out['D.logits'] = D(inp['d_img'], inp['d_c']).numpy()
        x, img, parsing = block(inp['b_x'], inp['b_img'], inp['b_ws'], None, {'32': inp['b_cat']}, noise_mode='const', fused_modconv=False)
        out['G.x'], out['G.img'], out['G.parsing'] = x.numpy(), img.numpy(), parsing.numpy()
        out['G_ema.y'] = spade(inp['s_x'], inp['s_feat']).numpy()
    snapshot = dict(training_set_kwargs=dict(class_name='training.dataset.UvitonDatasetFull', path='<synthetic>', use_labels=True),
                    G=block, D=D, G_ema=spade, augment_pipe=None)
    with open(os.path.join(GOLDEN, 'reference_snapshot.pkl'), 'wb') as f:
        pickle.dump(snapshot, f)
    blob = open(os.path.join(GOLDEN, 'reference_snapshot.pkl'), 'rb').read()
    assert b'def modulated_conv2d' not in blob and b'class Conv2dLayer' not in blob, 'reference source text leaked into the fixture'
    np.savez_compressed(os.path.join(GOLDEN, 'reference_snapshot_outputs.npz'), **out)
    print('snapshot fixture written:', len(blob), 'bytes')
