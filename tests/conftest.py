"""Test configuration: marker registration and import paths.

``-m "not gpu"`` : oracle vs golden fixtures, host logic, C-ABI symbol check (no GPU needed).
``-m gpu``       : parity of the HIP path against the oracle and the fixtures, through the C ABI.
"""

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'pasta-gan_amd')
for p in (PKG, ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel_err(a, b):
    """max |a-b| / (max |b| + tiny): the relative fp32 measure the parity bar is stated in."""
    a = torch.as_tensor(a).detach().to(torch.float64).cpu()
    b = torch.as_tensor(b).detach().to(torch.float64).cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(params=['f16x3', 'bf16x6'])
def arith(request):
    """Runs a parity test under the default three-product fp16 arithmetic AND under the six-product split-bf16 one, which stays
    held to the bounds it met before the default changed: a test takes ``arith`` and chooses its bound by it."""
    from torch_utils.ops import conv2d_gradfix
    old, conv2d_gradfix.conv_math = conv2d_gradfix.conv_math, request.param
    try:
        yield request.param
    finally:
        conv2d_gradfix.conv_math = old
