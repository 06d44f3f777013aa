"""The real generator / discriminator under DistributedDataParallel: two ranks sharing the one GPU of the test box
(gloo transport, device tensors), so that the native autograd Functions meet the flat gradient reducer's hooks and, in the
reference's arrangement, DDP's hooks, no_sync rounds and find_unused_parameters before the multi-GPU run on RCCL."""

import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out, ddp_mode):
    for p in (PKG, ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from training.training_loop_wo_flow_fullbody import TrainingStep, SyntheticFullBodyBatch, fashion_config
        torch.cuda.set_device(0)
        dev = torch.device('cuda', 0)
        cfg = fashion_config(channel_base=2048)
        step = TrainingStep(dev, cfg=cfg, num_gpus=world, rank=rank, batch_size=8 * world, batch_gpu=4, random_seed=0, ddp_mode=ddp_mode)
        data = SyntheticFullBodyBatch(8, dev, seed=rank)
        for _ in range(2):      # iteration 0 runs all four phases (incl. R1), two accumulation rounds each
            step.run(data)
        torch.cuda.synchronize()
        for name, p in list(step.G.named_parameters()) + list(step.D.named_parameters()):
            ref = p.detach().clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(ref, p.detach()), f'{name} diverged across ranks'
            assert torch.isfinite(p).all(), name
        moved = sum(float((a - b).abs().sum()) for a, b in zip(step.G.parameters(), step.G_ema.parameters()))
        assert moved > 0
        out.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        out.put((rank, 'FAIL: ' + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('ddp_mode', ['flat', 'torch'])
def test_generator_discriminator_ddp_two_ranks_one_gpu(ddp_mode):
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + (os.getpid() + (11 if ddp_mode == 'torch' else 0)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, ddp_mode)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == 'ok', f'rank {rank}: {msg}'


def _rccl_worker(port, out):
    """One rank on RCCL (torch.distributed backend 'nccl'): process-group creation bound to the device, barrier, broadcast of
    module states, and the flat reducer's hooks + asynchronous all-reduce on the real discriminator -- everything the N-rank run
    does except having a second GPU to talk to."""
    for p in (PKG, ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    try:
        torch.cuda.set_device(0)
        dev = torch.device('cuda', 0)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        assert dist.get_backend() == 'nccl'
        from training import networks
        from training.grad_reducer import FlatGradReducer, broadcast_module_states
        from training.training_loop_wo_flow_fullbody import fashion_config
        torch.manual_seed(0)
        kw = dict(fashion_config(channel_base=2048).D_kwargs)
        kw.pop('class_name')
        for key, value in dict(c_dim=512, img_resolution=256, img_channels=3).items():
            kw.setdefault(key, value)
        D = networks.Discriminator(**kw).to(dev)
        broadcast_module_states([D])
        dist.barrier()
        img = torch.randn([4, 3, 256, 256], device=dev)
        c = torch.randn([4, 512], device=dev)
        D(img, c).sum().backward()
        plain = [p.grad.clone() for p in D.parameters()]
        for p in D.parameters():
            p.grad = None
        # force_collective: the reducer's own all_reduce(bucket, async_op=True) + work.wait() run on RCCL's stream (a one-rank
        # sum is the identity, so the plain gradients must come back bit for bit)
        red = FlatGradReducer(D, world_size=1, bucket_mb=8, force_collective=True)
        assert len(red.buckets) > 1
        red.begin()
        red.arm(1)
        D(img, c).sum().backward()
        assert red.launched_early == len(red.buckets) and all(b.work is not None for b in red.buckets)
        red.finish()
        assert red.collectives_issued == len(red.buckets)
        torch.cuda.synchronize()
        for p, g in zip(D.parameters(), plain):
            assert p.grad is not None and torch.equal(p.grad, g)
        # a second phase on the same buckets (zeroed, re-armed) while the first phase's collectives have completed
        red.begin()
        red.arm(1)
        D(img, c).sum().backward()
        red.finish()
        torch.cuda.synchronize()
        for p, g in zip(D.parameters(), plain):
            assert torch.equal(p.grad, g)
        t = torch.ones([1 << 20], device=dev)
        dist.all_reduce(t)
        assert float(t.sum()) == float(1 << 20)
        out.put('ok')
    except Exception:  # noqa: BLE001
        import traceback
        out.put('FAIL: ' + traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_initialises_and_carries_the_flat_reducer_single_rank():
    """RCCL on this stack with world_size 1 (the test box has one GPU): the multi-GPU run's transport, hooks and stream
    semantics -- the bitwise gradients of a plain backward pass must come back from the buckets."""
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(31000 + os.getpid() % 2000, out))
    p.start()
    res = out.get(timeout=500)
    p.join(timeout=60)
    assert res == 'ok', res


def _bench_line(extra_env, *args):
    import json, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, **extra_env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1100)
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(2400)
def test_two_gpus_over_rccl_match_the_gloo_run():
    """BASELINE config 3 on the smallest node that can carry it: where the box shows two GPUs, ``bench.py --gpus 2``
    runs two ranks over RCCL -- (a) the line names the transport, (b) the replicas' parameters are bit-identical after the steps,
    (c) parameters and the last averaged gradients equal those of the same two ranks exchanging over gloo (a sum of two addends has one
    rounding whatever the transport; every kernel of the step is bitwise reproducible).  Skipped on the one-GPU boxes of this pool, where
    only the refusal (tests/test_bench_launch.py) and world size 1 (above) can run; the driver's 8-GPU node runs the curve itself."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs on one node')
    common = ('--gpus', '2', '--steps', '2', '--warmup', '1', '--batch-gpu', '4', '--no-cpu-baseline', '--no-variants', '--replica-check')
    rccl = _bench_line({'PASTA_DIST_BACKEND': 'nccl'}, *common)
    par = rccl['config']['parallelism']
    assert rccl['n_gpus'] == 2 and par.startswith('dp2') and 'backend nccl' in par and 'RCCL' in par and 'world_size 2' in par and 'REHEARSAL' not in par
    assert rccl['config']['global_batch'] == 8 and rccl['scaling'] == 'weak' and rccl['value'] > 0
    rc = rccl['replica_check']
    assert rc['world_size'] == 2 and rc['bit_identical'] and rc['grads'] > 0
    gloo = _bench_line({'PASTA_DIST_BACKEND': 'gloo'}, *common)
    gc = gloo['replica_check']
    assert gc['bit_identical'] and gc['grads'] == rc['grads']
    assert abs(rc['param_abs_sum'] / gc['param_abs_sum'] - 1) < 1e-6 and abs(rc['grad_abs_sum'] / gc['grad_abs_sum'] - 1) < 1e-6, (rc, gc)
