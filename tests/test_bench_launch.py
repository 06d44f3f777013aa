"""``python bench.py --gpus N`` as the driver calls it: the parent spawns the ranks itself (no torch.distributed.run
around it), relays rank 0's single JSON line, and fails when a rank fails."""

import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT


def _run(extra_env, *args, timeout=900):
    env = dict(os.environ, **extra_env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the failure path of a box without a GPU')
def test_launcher_reports_a_failed_rank():
    res = _run({}, '--gpus', '2', '--steps', '1', '--warmup', '1', '--no-cpu-baseline', timeout=300)
    assert res.returncode != 0
    assert 'exited with status' in res.stderr and res.stdout.strip() == ''


@pytest.mark.gpu
@pytest.mark.timeout(1200)
@pytest.mark.parametrize('ddp_mode', ['flat', 'torch'])
def test_two_ranks_on_one_gpu_over_gloo(ddp_mode):
    """Two ranks share the one card of the test box (gloo transport): the launcher, the process group, the gradient
    exchange and the max-over-ranks timing all run; the line says which transport carried the gradients."""
    res = _run({'PASTA_DIST_BACKEND': 'gloo'}, '--gpus', '2', '--steps', '1', '--warmup', '1', '--batch-gpu', '4',
               '--no-cpu-baseline', '--ddp-mode', ddp_mode, '--replica-check')
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 1 and out['value'] > 0
    assert out['config']['global_batch'] == 8
    par = out['config']['parallelism']
    assert par.startswith('dp2') and 'gloo' in par and 'world_size 2' in par and 'RCCL' not in par and ddp_mode in par
    assert out['roofline']['frac'] > 0
    rc = out['replica_check']       # the same check the two-GPU RCCL test makes (tests/test_ddp_gpu.py), here over the rehearsal transport
    assert rc['world_size'] == 2 and rc['bit_identical'] and rc['grads'] > 0 and rc['grad_abs_sum'] > 0


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_four_ranks_on_one_gpu_over_gloo():
    """Four ranks on the one card (the box admits six processes on it; eight cannot be rehearsed here -- the world-size-8 logic
    runs over gloo on the CPU, tests/test_ddp_cpu.py): 1/4 of the host cores per rank, four-way max-over-ranks timing, the
    flat reducer's buckets at world size 4 through the real generator / discriminator."""
    res = _run({'PASTA_DIST_BACKEND': 'gloo'}, '--gpus', '4', '--steps', '1', '--warmup', '1', '--batch-gpu', '2', '--no-cpu-baseline')
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 4 and out['config']['global_batch'] == 8 and out['value'] > 0 and out['scaling'] == 'weak'
    assert 'world_size 4' in out['config']['parallelism'] and 'REHEARSAL' in out['config']['parallelism']


@pytest.mark.gpu
def test_rccl_refuses_more_ranks_than_gpus():
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a single-GPU box')
    res = _run({'PASTA_DIST_BACKEND': 'nccl'}, '--gpus', '2', '--steps', '1', '--warmup', '1', '--no-cpu-baseline', timeout=600)
    assert res.returncode != 0 and 'needs 2 GPUs' in res.stderr


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_driver_style_launch_through_torch_distributed_run():
    """The driver's multi-GPU invocation: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`` -- every process is one rank (RANK / LOCAL_RANK / WORLD_SIZE from the launcher);
    rank 0 alone prints the line.  Two ranks on the one card of the test box, gloo transport."""
    env = dict(os.environ, PASTA_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    port = 23000 + os.getpid() % 2000
    res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1',
                          '--batch-gpu', '4', '--no-cpu-baseline'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 8 and out['scaling'] == 'weak'
    assert 'world_size 2' in out['config']['parallelism']
