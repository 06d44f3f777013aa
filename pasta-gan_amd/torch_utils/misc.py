"""Small helpers the layer code and the data-parallel step rely on.

API-compatible subset of the reference's torch_utils/misc.py: ``assert_shape`` (:80-95),
``profiled_function`` (:104-109), ``suppress_tracer_warnings`` (:67-71), ``nan_to_num`` (:45),
``params_and_buffers`` / ``named_params_and_buffers`` / ``copy_params_and_buffers`` (:151-166),
``ddp_sync`` (:172-179), ``check_ddp_consistency`` (:184-196) and the dataset sampler ``InfiniteSampler`` (:115-146), which
is how the reference's loop shards a dataset over the ranks (training_loop_wo_flow_fullbody.py:262-263); the synthetic batch
of the benchmark is sharded by seed instead.
"""

import contextlib
import functools
import itertools
import re
import warnings

import numpy as np
import torch

nan_to_num = torch.nan_to_num

def nan_to_num_(tensors, nan=0.0, posinf=None, neginf=None):
    """``torch.nan_to_num(t, nan, posinf, neginf, out=t)`` for every tensor of a list (the gradient clean-up before each
    optimiser step, training_loop_wo_flow_fullbody.py:513-515).  Contiguous float32 GPU tensors go through ONE launch of
    ``pasta_nan_to_num_multi`` per 96 tensors; anything else (CPU tensors of the gloo tests, other dtypes) is done one by one."""
    import ctypes
    fast = [t for t in tensors if t.device.type == 'cuda' and t.dtype == torch.float32 and t.is_contiguous()]
    rest = [t for t in tensors if not (t.device.type == 'cuda' and t.dtype == torch.float32 and t.is_contiguous())]
    for t in rest:
        torch.nan_to_num(t, nan=nan, posinf=posinf, neginf=neginf, out=t)
    by_device = {}
    for t in fast:
        by_device.setdefault(t.device, []).append(t)
    for device, ts in by_device.items():
        from .ops import _native
        fmax = float(torch.finfo(torch.float32).max)
        ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        numels = (ctypes.c_int64 * len(ts))(*[t.numel() for t in ts])
        with torch.cuda.device(device):
            st = _native.lib().pasta_nan_to_num_multi(ptrs, numels, len(ts), float(nan), fmax if posinf is None else float(posinf),
                                                      -fmax if neginf is None else float(neginf), _native.stream())
        _native.check(st)

@contextlib.contextmanager
def suppress_tracer_warnings():
    """Silence torch.jit.TracerWarning inside the block."""
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=torch.jit.TracerWarning)
        yield

def assert_shape(tensor, ref_shape):
    """AssertionError unless ``tensor`` has the rank of ``ref_shape`` and every size that ``ref_shape`` names (None = any)."""
    got = tuple(int(s) for s in tensor.shape)
    if len(got) != len(ref_shape):
        raise AssertionError(f'Wrong number of dimensions: got {len(got)}, expected {len(ref_shape)}')
    bad = [(d, g, int(r)) for d, (g, r) in enumerate(zip(got, ref_shape)) if r is not None and g != int(r)]
    if bad:
        d, g, r = bad[0]
        raise AssertionError(f'Wrong size for dimension {d}: got {g}, expected {r}')

def profiled_function(fn):
    """Decorator: run ``fn`` inside a ``record_function`` range carrying its name (visible in torch.profiler traces)."""
    @functools.wraps(fn)
    def ranged(*args, **kwargs):
        with torch.autograd.profiler.record_function(fn.__name__):
            return fn(*args, **kwargs)
    return ranged

#----------------------------------------------------------------------------

def named_params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return [*module.named_parameters(), *module.named_buffers()]

def params_and_buffers(module):
    return [t for _name, t in named_params_and_buffers(module)]

def copy_params_and_buffers(src_module, dst_module, require_all=False):
    """Overwrite ``dst_module``'s tensors with the same-named tensors of ``src_module`` (``requires_grad`` flags stay);
    ``require_all``: a destination name missing from the source is an error."""
    source = dict(named_params_and_buffers(src_module))
    for name, dst in named_params_and_buffers(dst_module):
        if name not in source:
            assert not require_all, f'{name} missing from the source module'
            continue
        keep = dst.requires_grad
        dst.copy_(source[name].detach()).requires_grad_(keep)

#----------------------------------------------------------------------------

@contextlib.contextmanager
def ddp_sync(module, sync):
    """``with ddp_sync(m, sync):`` -- gradients produced inside are all-reduced by a DistributedDataParallel ``m`` only
    when ``sync``; any other module (single GPU, or the flat gradient reducer, which is gated by the step) passes through."""
    assert isinstance(module, torch.nn.Module)
    wrapped = isinstance(module, torch.nn.parallel.DistributedDataParallel)
    with (module.no_sync() if wrapped and not sync else contextlib.nullcontext()):
        yield

def check_ddp_consistency(module, ignore_regex=None):
    """Every parameter / buffer must equal rank 0's copy bit for bit (NaNs compare equal); names matching
    ``ignore_regex`` (as ``<ClassName>.<tensor name>``) are exempt."""
    assert isinstance(module, torch.nn.Module)
    prefix = type(module).__name__ + '.'
    for name, tensor in named_params_and_buffers(module):
        if ignore_regex is not None and re.fullmatch(ignore_regex, prefix + name):
            continue
        mine = nan_to_num(tensor.detach())
        theirs = mine.clone()
        torch.distributed.broadcast(tensor=theirs, src=0)
        assert torch.equal(mine, theirs), prefix + name

#----------------------------------------------------------------------------


class InfiniteSampler(torch.utils.data.Sampler):
    """Endless index stream over ``dataset`` for ``torch.utils.data.DataLoader(sampler=...)``: position ``t`` of ONE global
    stream (identical on every rank: same seed, same draws) belongs to rank ``t % num_replicas``, so the ranks read disjoint
    items; after each position the item just passed is swapped with one at most ``window_size * len(dataset)`` places behind
    it, which keeps the order drifting without ever reshuffling the whole set.  Same constructor, same index stream for a
    given seed as the reference's class (misc.py:115-146; pinned by tests/golden/sampler.npz)."""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        if len(dataset) <= 0:
            raise AssertionError('InfiniteSampler: empty dataset')
        if not (num_replicas > 0 and 0 <= rank < num_replicas):
            raise AssertionError(f'InfiniteSampler: rank {rank} is not in [0, {num_replicas})')
        if not 0 <= window_size <= 1:
            raise AssertionError('InfiniteSampler: window_size must lie in [0, 1]')
        super().__init__()
        self.dataset, self.rank, self.num_replicas = dataset, rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size

    def __iter__(self):
        n = len(self.dataset)
        perm = np.arange(n)
        draw, reach = None, 0
        if self.shuffle:
            draw = np.random.RandomState(self.seed)
            draw.shuffle(perm)
            reach = int(np.rint(n * self.window_size))
        for t in itertools.count():
            here = t % n
            if t % self.num_replicas == self.rank:
                yield perm[here]
            if reach >= 2:      # every rank makes every draw, its own positions or not: the stream stays global
                back = (here - draw.randint(reach)) % n
                perm[[here, back]] = perm[[back, here]]
