"""Small helpers the layer code and the data-parallel step rely on.

API-compatible subset of the reference's torch_utils/misc.py: ``assert_shape`` (:80-95),
``profiled_function`` (:104-109), ``suppress_tracer_warnings`` (:67-71), ``nan_to_num`` (:45),
``params_and_buffers`` / ``named_params_and_buffers`` / ``copy_params_and_buffers`` (:151-166),
``ddp_sync`` (:172-179), ``check_ddp_consistency`` (:184-196) and ``InfiniteSampler`` (:115-146).
"""

import contextlib
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

class suppress_tracer_warnings(warnings.catch_warnings):
    def __enter__(self):
        super().__enter__()
        warnings.simplefilter('ignore', category=torch.jit.TracerWarning)
        return self

def assert_shape(tensor, ref_shape):
    """Raise AssertionError unless ``tensor.shape`` matches ``ref_shape`` (``None`` = any size)."""
    if tensor.ndim != len(ref_shape):
        raise AssertionError(f'Wrong number of dimensions: got {tensor.ndim}, expected {len(ref_shape)}')
    for idx, (size, ref_size) in enumerate(zip(tensor.shape, ref_shape)):
        if ref_size is not None and int(size) != int(ref_size):
            raise AssertionError(f'Wrong size for dimension {idx}: got {size}, expected {ref_size}')

def profiled_function(fn):
    """Wrap ``fn`` in a ``record_function`` scope named after it (shows up in torch.profiler traces)."""
    def decorator(*args, **kwargs):
        with torch.autograd.profiler.record_function(fn.__name__):
            return fn(*args, **kwargs)
    decorator.__name__ = fn.__name__
    return decorator

#----------------------------------------------------------------------------

def params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.parameters()) + list(module.buffers())

def named_params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.named_parameters()) + list(module.named_buffers())

def copy_params_and_buffers(src_module, dst_module, require_all=False):
    """Copy same-named parameters and buffers; with ``require_all`` every destination name must exist in the source."""
    src = dict(named_params_and_buffers(src_module))
    for name, tensor in named_params_and_buffers(dst_module):
        assert (name in src) or (not require_all), name
        if name in src:
            tensor.copy_(src[name].detach()).requires_grad_(tensor.requires_grad)

#----------------------------------------------------------------------------

@contextlib.contextmanager
def ddp_sync(module, sync):
    """Run the block with DistributedDataParallel gradient all-reduce on (``sync``) or suppressed."""
    assert isinstance(module, torch.nn.Module)
    if sync or not isinstance(module, torch.nn.parallel.DistributedDataParallel):
        yield
    else:
        with module.no_sync():
            yield

def check_ddp_consistency(module, ignore_regex=None):
    """Assert that every parameter / buffer equals rank 0's copy (broadcast + compare)."""
    assert isinstance(module, torch.nn.Module)
    for name, tensor in named_params_and_buffers(module):
        fullname = type(module).__name__ + '.' + name
        if ignore_regex is not None and re.fullmatch(ignore_regex, fullname):
            continue
        tensor = tensor.detach()
        other = tensor.clone()
        torch.distributed.broadcast(tensor=other, src=0)
        assert (nan_to_num(tensor) == nan_to_num(other)).all(), fullname

#----------------------------------------------------------------------------

class InfiniteSampler(torch.utils.data.Sampler):
    """Endless index stream, sharded ``rank::num_replicas``, with a sliding-window shuffle."""
    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        assert len(dataset) > 0 and num_replicas > 0 and 0 <= rank < num_replicas and 0 <= window_size <= 1
        super().__init__()
        self.dataset, self.rank, self.num_replicas = dataset, rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size

    def __iter__(self):
        order = np.arange(len(self.dataset))
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.RandomState(self.seed)
            rnd.shuffle(order)
            window = int(np.rint(order.size * self.window_size))
        idx = 0
        while True:
            i = idx % order.size
            if idx % self.num_replicas == self.rank:
                yield order[i]
            if window >= 2:
                j = (i - rnd.randint(window)) % order.size
                order[i], order[j] = order[j], order[i]
            idx += 1
