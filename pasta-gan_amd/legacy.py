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

_FP16_OVERRIDES = dict(num_fp16_res=4, conv_clamp=256)     # what the reference's loader forces (legacy.py:45-59)

def _with_fp16_blocks(net, generator):
    """``net`` rebuilt from its recorded constructor arguments with the four highest resolutions in fp16 and the matching
    clamp (a generator keeps them under ``synthesis_kwargs``); the original is returned when nothing changes."""
    want = copy.deepcopy(net.init_kwargs)
    target = want
    if generator:
        target = want['synthesis_kwargs'] = dnnlib.EasyDict(want.get('synthesis_kwargs') or {})
    target.update(_FP16_OVERRIDES)
    if want == net.init_kwargs:
        return net
    rebuilt = type(net)(**want).eval().requires_grad_(False)
    misc.copy_params_and_buffers(net, rebuilt, require_all=True)
    return rebuilt

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

    if force_fp16:
        data.update({key: _with_fp16_blocks(data[key], generator=(key != 'D')) for key in ('G', 'D', 'G_ema')})
    return data

#----------------------------------------------------------------------------
