"""The oracle (oracle/ref_ops.py) against fixtures produced by the reference itself."""

import json

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import ref_ops as R

TOL = 2e-6   # fp32 rounding of a different summation order


def t(a, grad=False):
    x = torch.from_numpy(np.asarray(a)).clone()
    return x.requires_grad_(True) if grad else x


def _cases(fname):
    g = load_golden(fname)
    return g, json.loads(str(g['manifest']))


def test_setup_filter():
    g, specs = _cases('ops_setup_filter.npz')
    for i, s in enumerate(specs):
        got = R.setup_filter(**s)
        assert rel_err(got, g[f'f{i}']) < 1e-7, s


@pytest.mark.parametrize('idx', range(18))
def test_upfirdn2d(idx):
    g, cases = _cases('ops_upfirdn2d.npz')
    c = cases[idx]
    n = c['name']
    x = t(g[n + '.x'], True)
    f = t(g[n + '.f']) if c['f'] is not None else None
    y = R.upfirdn2d(x, f, **c['call'])
    assert rel_err(y, g[n + '.y']) < TOL
    dx, = torch.autograd.grad(y, x, t(g[n + '.dy']))
    assert rel_err(dx, g[n + '.dx']) < TOL


def test_upfirdn2d_case_count():
    assert len(_cases('ops_upfirdn2d.npz')[1]) == 18


@pytest.mark.parametrize('idx', range(24))
def test_bias_act(idx):
    g, cases = _cases('ops_bias_act.npz')
    c = cases[idx]
    n = c['name']
    x = t(g[n + '.x'], True)
    b = t(g[n + '.b'], True) if c['bias'] else None
    dy = t(g[n + '.dy'], True)
    y = R.bias_act(x, b, act=c['act'], **c['kw'])
    assert rel_err(y, g[n + '.y']) < TOL
    grads = torch.autograd.grad(y, [x] + ([b] if b is not None else []), dy, create_graph=True)
    assert rel_err(grads[0], g[n + '.dx']) < TOL
    if b is not None:
        assert rel_err(grads[1], g[n + '.db']) < 1e-5
    gg = torch.autograd.grad(grads[0], [dy, x], t(g[n + '.ddx']), allow_unused=True)
    assert rel_err(gg[0], g[n + '.g_dy']) < TOL
    if gg[1] is not None:
        assert rel_err(gg[1], g[n + '.g_x']) < 1e-5


def test_bias_act_case_count():
    assert len(_cases('ops_bias_act.npz')[1]) == 24


@pytest.mark.parametrize('fast', [True, False])
@pytest.mark.parametrize('idx', range(15))
def test_conv2d_resample(idx, fast):
    g, cases = _cases('ops_conv2d_resample.npz')
    c = cases[idx]
    n = c['name']
    x, w = t(g[n + '.x'], True), t(g[n + '.w'], True)
    f = R.setup_filter(c['f']) if c.get('f') is not None else None
    y = R.conv2d_resample(x, w, f=f, fast=fast, **c['kw'])
    assert rel_err(y, g[n + '.y']) < 1e-5
    dx, dw = torch.autograd.grad(y, [x, w], t(g[n + '.dy']))
    assert rel_err(dx, g[n + '.dx']) < 1e-5
    assert rel_err(dw, g[n + '.dw']) < 1e-5


def test_fma():
    g = load_golden('ops_fma.npz')
    a, b, c = t(g['a'], True), t(g['b'], True), t(g['c'], True)
    y = R.fma(a, b, c)
    assert rel_err(y, g['y']) < 1e-6
    da, db, dc = torch.autograd.grad(y, [a, b, c], t(g['dy']))
    assert rel_err(da, g['da']) < 1e-6 and rel_err(db, g['db']) < 1e-6 and rel_err(dc, g['dc']) < 1e-6
