"""Snapshot loader with the reference's entry point: ``legacy.load_network_pkl(f)`` -> dict with ``G``, ``D``, ``G_ema``,
``training_set_kwargs``, ``augment_pipe`` (reference legacy.py:20-60; used by test.py:92-93 and test_512.py:92-93).

Network classes in the pickle are re-bound by name to this package's ``training.networks`` classes
(torch_utils/persistence.py), so a snapshot written by the reference loads without executing the reference's
``networks.py`` -- which cannot be imported on ROCm -- and runs on the HIP operators.  The TensorFlow-era pickles that the
reference also converts (legacy.py:64-320) predate PASTA-GAN and are rejected."""

import copy
import pickle

import torch

import dnnlib
from torch_utils import misc
import training.networks  # noqa: F401  (registers the persistent classes that pickled names are bound to)

#----------------------------------------------------------------------------

class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == 'dnnlib.tflib.network' and name == 'Network':
            raise NotImplementedError('TensorFlow StyleGAN pickles are not supported; convert them with the reference repository first')
        return super().find_class(module, name)

def load_network_pkl(f, force_fp16=False):
    data = _Unpickler(f).load()
    if not isinstance(data, dict):
        raise ValueError('not a PASTA-GAN / StyleGAN2-ADA PyTorch snapshot (expected a dict with G, D, G_ema)')
    data.setdefault('training_set_kwargs', None)
    data.setdefault('augment_pipe', None)
    for key in ['G', 'D', 'G_ema']:
        assert isinstance(data[key], torch.nn.Module), f'snapshot entry {key!r} is not a module'
    assert isinstance(data['training_set_kwargs'], (dict, type(None)))
    assert isinstance(data['augment_pipe'], (torch.nn.Module, type(None)))

    if force_fp16:      # rebuild with fp16 enabled in the four highest resolutions, as legacy.py:45-59
        for key in ['G', 'D', 'G_ema']:
            old = data[key]
            kwargs = copy.deepcopy(old.init_kwargs)
            if key.startswith('G'):
                kwargs.synthesis_kwargs = dnnlib.EasyDict(kwargs.get('synthesis_kwargs', {}))
                kwargs.synthesis_kwargs.num_fp16_res = 4
                kwargs.synthesis_kwargs.conv_clamp = 256
            else:
                kwargs.num_fp16_res = 4
                kwargs.conv_clamp = 256
            if kwargs != old.init_kwargs:
                new = type(old)(**kwargs).eval().requires_grad_(False)
                misc.copy_params_and_buffers(old, new, require_all=True)
                data[key] = new
    return data

#----------------------------------------------------------------------------
