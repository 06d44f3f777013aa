"""The data-parallel step at world size 2 over gloo (CPU), in both exchange modes (the flat gradient reducer and the
reference's DistributedDataParallel wrappers): gradients are averaged across ranks exactly once per phase, replicas stay
identical, parameters no backward reaches keep ``grad = None``, and the style-encoder-only Greg phase wedges nothing."""

import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG


def _worker(rank, world, port, out, ddp_mode):
    for p in (PKG, ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import dnnlib
        from training.training_loop_wo_flow_fullbody import TrainingStep, SyntheticFullBodyBatch, fashion_config
        torch.set_num_threads(1)
        cfg = fashion_config()
        cfg.G_kwargs = dnnlib.EasyDict(class_name='tiny_models.TinyG')
        cfg.D_kwargs = dnnlib.EasyDict(class_name='tiny_models.TinyD')
        cfg.loss_kwargs.style_mixing_prob = 0
        dev = torch.device('cpu')
        step = TrainingStep(dev, cfg=cfg, num_gpus=world, rank=rank, batch_size=4 * world, batch_gpu=2, random_seed=0, ddp_mode=ddp_mode)
        data = SyntheticFullBodyBatch(4, dev, seed=rank, res=16)

        # replicas start identical (DDP constructor broadcast from rank 0), although seeds differ per rank
        for name, p in list(step.G.named_parameters()) + list(step.D.named_parameters()):
            ref = p.detach().clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(ref, p.detach()), f'{name} differs after construction'

        # expected D gradient of the Dmain phase = mean over ranks of each rank's two-round accumulated gradient
        import copy
        from training.loss_wo_flow_fullbody import StyleGAN2Loss
        Gc, Dc = copy.deepcopy(step.G), copy.deepcopy(step.D)
        local = StyleGAN2Loss(device=dev, G_mapping=Gc.mapping, G_synthesis=Gc.synthesis, G_const_encoding=Gc.const_encoding,
                              G_style_encoding=Gc.style_encoding, D=Dc, **{k: v for k, v in cfg.loss_kwargs.items() if k != 'class_name'})
        Dc.requires_grad_(True)
        for r in data.split(2):
            local.accumulate_gradients(phase='Dmain', gen_z=torch.zeros([2, 0]), sync=True, gain=1, **r)
        expected = []
        for p in Dc.parameters():
            g = p.grad.clone()
            dist.all_reduce(g)
            expected.append(g / world)

        # the same phase through the step's exchange (rounds: nothing sent on the first, all-reduce on the last)
        step.D.requires_grad_(True)
        red = step.reducers.get('D')
        assert (red is not None) == (ddp_mode == 'flat')
        if red is not None:
            red.begin()
        for i, r in enumerate(data.split(2)):
            if red is not None and i == 1:
                red.arm(step.loss.backward_passes('Dmain'))
            step.loss.accumulate_gradients(phase='Dmain', gen_z=torch.zeros([2, 0]), sync=(i == 1), gain=1, **r)
        if red is not None:
            red.finish()
            assert red.launched_early == len(red.buckets)       # every bucket left during backward, none waited for finish()
        for p, e in zip(step.D.parameters(), expected):
            assert torch.allclose(p.grad, e, rtol=1e-5, atol=1e-7)
        step.D.requires_grad_(False)
        for p in step.D.parameters():
            p.grad = None

        # full iterations, including Greg (forward of the style encoder without a backward) and Dreg (R1)
        for _ in range(5):
            step.run(data)
        for name, p in list(step.G.named_parameters()) + list(step.D.named_parameters()):
            ref = p.detach().clone()
            dist.broadcast(ref, src=0)
            assert torch.equal(ref, p.detach()), f'{name} diverged across ranks'
        from torch_utils import misc
        misc.check_ddp_consistency(step.G, ignore_regex=r'.*\.w_avg')
        # a parameter no backward reaches is skipped by Adam (grad None), as under the reference's find_unused_parameters
        assert step.G.synthesis.unused.grad is None and float(step.G.synthesis.unused.abs().sum()) == 0
        out.put((rank, 'ok'))
    except Exception as e:  # noqa: BLE001
        import traceback
        out.put((rank, 'FAIL: ' + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _spawn(world, ddp_mode, salt):
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + (os.getpid() + salt) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, ddp_mode)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == 'ok', f'rank {rank}: {msg}'


@pytest.mark.timeout(600)
def test_data_parallel_step_world_size_8():
    """The world size of BASELINE config 3 (one node, 8 ranks), rehearsed over gloo on the CPU -- the GPU box admits at most six
    processes on its one card, so eight ranks cannot share it: bucket filling / arming, the 1/8 pre-scale, unused parameters and
    replica consistency over five iterations with lazy regularisation, at the real rank count."""
    _spawn(8, 'flat', 13)


@pytest.mark.timeout(300)
@pytest.mark.parametrize('ddp_mode', ['flat', 'torch'])
def test_data_parallel_step_world_size_2(ddp_mode):
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + (os.getpid() + (7 if ddp_mode == 'torch' else 0)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, ddp_mode)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == 'ok', f'rank {rank}: {msg}'


def test_flat_reducer_refuses_a_gradient_after_its_bucket_left(tmp_path):
    """arm() told one backward pass, the loss makes two -> the second pass's gradients would be added to a
    bucket that is already on the wire (pre-scaled).  Where an exchange happens the reducer must raise, not diverge silently;
    where none happens (one rank, no forced collective) a late gradient is harmless and is accepted, and a
    backward pass after finish() -- outside the begin / arm / finish protocol -- is not "late"."""
    import torch.distributed as dist
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from training.grad_reducer import FlatGradReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
    x = torch.randn(3, 8)
    # one rank, no exchange: nothing to protect
    red = FlatGradReducer(net, world_size=1)
    red.begin()
    red.arm(2)
    net(x).sum().backward()
    net(x).sum().backward()           # two passes announced, two made: fine
    red.finish()
    assert red.launched_early == len(red.buckets)
    red.begin()
    red.arm(1)
    net(x).sum().backward()
    once = net[0].weight.grad.clone()
    net(x).sum().backward()           # a second pass nobody announced: accumulated like any gradient
    assert torch.allclose(net[0].weight.grad, 2 * once)
    red.finish()
    net(x).sum().backward()           # after the phase: not the reducer's business
    red.remove()
    # the same with the collective in force (one-rank gloo group): the late gradient raises
    dist.init_process_group('gloo', init_method='file://' + str(tmp_path / 'store'), rank=0, world_size=1)
    try:
        red = FlatGradReducer(net, world_size=1, force_collective=True)
        red.begin()
        red.arm(1)
        net(x).sum().backward()
        with pytest.raises(RuntimeError, match='after its bucket was sent'):
            net(x).sum().backward()
        red.finish()
        net(x).sum().backward()       # launched flags are cleared when the phase ends
        red.remove()
    finally:
        dist.destroy_process_group()
