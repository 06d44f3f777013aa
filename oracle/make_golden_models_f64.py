"""The reference's GeneratorFull evaluated in DOUBLE precision on the inputs and weights of ``models_fullbody.npz`` -- build
container only, called by ``oracle/make_golden.py --only models_f64``.  TEST INFRASTRUCTURE.

Why: some parameter gradients of the closed-form-filled generator are what a cancellation leaves over (the demodulated style
path: ``synthesis.b64.conv0.affine.weight`` has entries of 1e-7 against 1e-3 elsewhere), and the reference's own fp32 evaluation
is 1.1e-3 away from its exact-arithmetic value there (measured: fp32 CPU reference, fp32 MFMA and split-bf16 all land 1.1 - 1.5e-3
from the fp64 value, the three-product fp16 arithmetic 1.5e-4).  A parity bar of 1e-3 against ONE fp32 rounding of such a quantity
measures luck; this fixture pins what the reference's ALGORITHM computes, so that a result may meet the bar against either
evaluation of the reference (tests/test_models_gpu.py).
"""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, os.path.dirname(HERE))
from oracle import param_fill as PF  # noqa: E402
from oracle.make_golden_models import GRAD_KEYS_G  # noqa: E402


def gen_models_f64(ref_root, import_reference_networks):
    rn = import_reference_networks(ref_root)
    torch.manual_seed(0)
    inp = {k: (v.double() if v.is_floating_point() else v) for k, v in PF.make_inputs(n=2, seed=0).items()}
    G = PF.fill_module(rn.GeneratorFull(**PF.G_KWARGS)).double().train().requires_grad_(True)
    # Harness-side accommodation: the reference writes `torch.float32` wherever it fixes its working precision (block inputs
    # networks.py:5684, mapping :233, filters conv2d_resample.py:86).  For this one forward / backward the NAME torch.float32 is bound
    # to the double type, so the reference's own code runs every one of those casts -- and hence the whole network -- in fp64.
    real_f32 = torch.float32
    torch.float32 = torch.float64
    try:
        img, fin, par = G(inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
                          inp['denorm_upper_mask'], inp['denorm_lower_mask'], noise_mode='const')
        assert img.dtype == torch.float64
        probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
        probe.backward()
    finally:
        torch.float32 = real_f32
    out = {'G.probe': np.array([probe.item()])}
    sd = dict(G.named_parameters())
    for k in GRAD_KEYS_G:
        flat = sd[k].grad.detach().reshape(-1)
        step = max(flat.numel() // 4096, 1)
        out['G.grad.' + k + '.sample'] = flat[::step][:4096].numpy().copy()           # float64, the sampling of param_fill.summarize
    flat = img.detach().reshape(-1)
    out['G.img.sample'] = flat[::max(flat.numel() // 4096, 1)][:4096].numpy().copy()
    np.savez_compressed(os.path.join(GOLDEN, 'models_fullbody_f64.npz'), **out)
    print('fp64 model fixture written:', len(out), 'arrays')
