"""Build and load the native HIP library behind ``torch_utils.ops``.

Stands where the reference's JIT plugin loader stands (torch_utils/custom_ops.py:46-124,
``get_plugin(module_name, sources, **build_kwargs)``), but for one ahead-of-time library:
``csrc/*.hip`` -> ``hipcc --offload-arch=gfx950`` -> ``lib/libpasta_hip.so`` kept in-tree,
loaded with ``ctypes`` through the C ABI declared in ``include/pasta_hip.h``.

There is no fallback: if the library cannot be built or loaded, every op raises.
"""

import ctypes
import fcntl
import hashlib
import os
import shutil
import subprocess
import threading
import time
import warnings

# PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so) and puts it in the process-global symbol scope.
# Importing torch BEFORE loading libpasta_hip.so makes the library's HIP calls bind to that same runtime, so its
# kernels and PyTorch's share devices, streams and allocations. Loaded the other way round the process ends up with
# two HIP runtimes and the second one finds no device.
import torch  # noqa: F401  (must precede ctypes.CDLL below)

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # .../pasta-gan_amd
_CSRC = os.path.join(_ROOT, 'csrc')
_LIBDIR = os.path.join(_ROOT, 'lib')
_OBJDIR = os.path.join(_ROOT, 'build')
_INCLUDE = os.path.join(os.path.dirname(_ROOT), 'include')
LIB_NAME = 'libpasta_hip.so'
EXPECTED_ABI = 21                   # PASTA_ABI_VERSION of include/pasta_hip.h = pasta_abi_version() of csrc/common.hip
ARCH = 'gfx950'

_lock = threading.Lock()
_cached_plugins = dict()

#----------------------------------------------------------------------------

def _sources():
    return sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith('.hip'))

def _digest(sources, extra_flags=()):
    h = hashlib.md5()
    # the compile flags are part of what the library IS: a -DPASTA_ABLATE=... build must never carry the default stamp
    h.update(repr([str(f) for f in extra_flags]).encode())
    extra = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith('.h')]
    extra.append(os.path.join(_INCLUDE, 'pasta_hip.h'))
    for path in list(sources) + extra:
        h.update(os.path.basename(path).encode())
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()

def _hipcc():
    return shutil.which('hipcc') or ('/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else None)

def _stamp_matches(lib_path, stamp, digest):
    if not (os.path.exists(lib_path) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        return f.read().strip() == digest

def build(force=False, verbose=False, extra_flags=()):
    """Compile every ``csrc/*.hip`` for gfx950 and link ``lib/libpasta_hip.so``.

    Skips the work when the stored source digest matches. Returns the library path.  Concurrent callers (several ranks
    importing the ops at once) are serialised by an exclusive lock on ``lib/.build.lock`` -- the reference loader does
    the same with a file baton (custom_ops.py:86-92) -- and whoever gets the lock second finds the stamp up to date."""
    sources = _sources()
    lib_path = os.path.join(_LIBDIR, LIB_NAME)
    stamp = lib_path + '.md5'
    digest = _digest(sources, extra_flags)
    if not force and _stamp_matches(lib_path, stamp, digest):
        return lib_path
    hipcc = _hipcc()
    if hipcc is None:
        raise RuntimeError('hipcc not found: cannot build ' + LIB_NAME)
    os.makedirs(_LIBDIR, exist_ok=True)
    os.makedirs(_OBJDIR, exist_ok=True)
    with open(os.path.join(_LIBDIR, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _stamp_matches(lib_path, stamp, digest):
                return lib_path
            return _compile_and_link(hipcc, sources, lib_path, stamp, digest, verbose, extra_flags)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)

def _unit_weight(src):
    """Rough compile cost of a translation unit, for the build order: the kernel instances are in the conv_tu_* units and in bias_act."""
    name = os.path.basename(src)
    return (4 if name.startswith(('conv_tu_fwd_rows', 'conv_tu_rows2d', 'conv_tu_fwd_base', 'bias_act')) else 2 if name.startswith(('conv_tu_', 'upfirdn2d')) else 1)

def _compile_and_link(hipcc, sources, lib_path, stamp, digest, verbose, extra_flags):
    flags = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast', '-I', _INCLUDE] + list(extra_flags)
    # one hipcc per translation unit, at most one per core at a time, the large units first (the convolution family is thirteen
    # units, csrc/conv_launch.h: a forced build takes about a minute on eight cores)
    objs = [os.path.join(_OBJDIR, os.path.basename(src)[:-4] + '.o') for src in sources]
    queue = sorted(zip(sources, objs), key=lambda so: -_unit_weight(so[0]))
    jobs = max(1, min(len(queue), int(os.environ.get('PASTA_BUILD_JOBS', 0)) or os.cpu_count() or 1))
    running, failed = [], None
    while (queue and failed is None) or running:
        while queue and failed is None and len(running) < jobs:
            src, obj = queue.pop(0)
            cmd = [hipcc] + flags + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            log = open(obj[:-2] + '.log', 'wb')      # a file, not a pipe: a unit that prints many warnings must not block on a full pipe
            running.append((src, log, subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT)))
        for item in list(running):
            src, log, proc = item
            if proc.poll() is None:
                continue
            running.remove(item)
            log.close()
            if proc.returncode != 0 and failed is None:
                with open(log.name, 'rb') as f:
                    failed = 'hipcc failed on %s:\n%s' % (src, f.read().decode(errors='replace'))
        if running:
            time.sleep(0.05)
    if failed is not None:
        raise RuntimeError(failed)
    tmp = lib_path + '.tmp%d' % os.getpid()
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', tmp] + objs
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError('link failed:\n' + res.stdout.decode(errors='replace'))
    os.replace(tmp, lib_path)
    with open(stamp, 'w') as f:
        f.write(digest)
    return lib_path

#----------------------------------------------------------------------------

_c_i32 = ctypes.c_int32
_c_i64 = ctypes.c_int64
_c_f32 = ctypes.c_float
_c_ptr = ctypes.c_void_p

class ConvDesc(ctypes.Structure):
    """Mirror of ``pasta_conv_desc`` (include/pasta_hip.h)."""
    _fields_ = [(name, _c_i32) for name in (
        'N', 'C_in', 'H', 'W', 'C_out', 'OH', 'OW', 'kh', 'kw', 'stride',
        'pad_h', 'pad_w', 'groups', 'transposed', 'flip', 'math')] + [('wscale', _c_f32), ('io_dtype', _c_i32),
                                                                      ('x_amax', _c_ptr), ('x2', _c_ptr), ('x2_amax', _c_ptr), ('C1', _c_i32), ('dy_amax', _c_ptr), ('x_layout', _c_i32), ('w_prepacked', _c_i32)]

class ConvEpilogue(ctypes.Structure):
    """Mirror of ``pasta_conv_epilogue`` (include/pasta_hip.h)."""
    _fields_ = [('bias', _c_ptr), ('act', _c_i32), ('alpha', _c_f32), ('gain', _c_f32), ('clamp', _c_f32), ('res', _c_ptr),
                ('noise', _c_ptr), ('noise_strength', _c_ptr), ('noise_per_sample', _c_i32), ('y_amax', _c_ptr)]

class AdaConfig(ctypes.Structure):
    """Mirror of ``pasta_ada_config`` (include/pasta_hip.h)."""
    _fields_ = [(name, _c_f32) for name in (
        'xflip', 'rotate90', 'xint', 'xint_max', 'scale', 'rotate', 'aniso', 'xfrac', 'scale_std', 'rotate_max', 'aniso_std',
        'xfrac_std', 'brightness', 'contrast', 'lumaflip', 'hue', 'saturation', 'brightness_std', 'contrast_std', 'hue_max',
        'saturation_std')]

# name -> (restype, argtypes); exactly the symbols include/pasta_hip.h declares.
ABI = {
    'pasta_last_error':   (ctypes.c_char_p, []),
    'pasta_abi_version':  (ctypes.c_int, []),
    'pasta_build_info':   (ctypes.c_char_p, []),
    'pasta_upfirdn2d':    (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, ctypes.c_int,
                                          ctypes.POINTER(_c_i32), ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i32),
                                          ctypes.POINTER(_c_i32), ctypes.POINTER(_c_i64)] + [ctypes.c_int] * 9 + [_c_f32, _c_ptr, _c_ptr, _c_ptr]),
    'pasta_bias_act':     (ctypes.c_int, [_c_ptr] * 6 + [ctypes.c_int, _c_i64, ctypes.c_int, _c_i64, ctypes.c_int, ctypes.c_int,
                                                           _c_f32, _c_f32, _c_f32, _c_ptr, _c_ptr]),
    'pasta_bias_grad_workspace': (_c_i64, [_c_i64, ctypes.c_int, _c_i64]),
    'pasta_bias_grad':    (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, ctypes.c_int, _c_i64, ctypes.c_int, _c_i64, _c_ptr]),
    'pasta_bias_act_grad_db_workspace': (_c_i64, [ctypes.c_int, _c_i64, ctypes.c_int, _c_i64, ctypes.c_int]),
    'pasta_bias_act_grad_db': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, ctypes.c_int, _c_i64, ctypes.c_int, _c_i64, ctypes.c_int,
                                              ctypes.c_float, ctypes.c_float, ctypes.c_float, _c_ptr, _c_ptr]),
    'pasta_pieces_bytes':  (_c_i64, [ctypes.c_int] * 4),
    'pasta_blur_pieces':   (ctypes.c_int, [_c_ptr] * 5 + [ctypes.c_int] * 9 + [_c_f32, _c_ptr]),
    'pasta_pieces_unpack': (ctypes.c_int, [_c_ptr] * 3 + [ctypes.c_int] * 4 + [_c_ptr]),
    'pasta_pieces_pack': (ctypes.c_int, [_c_ptr] * 3 + [ctypes.c_int] * 4 + [_c_ptr]),
    'pasta_conv2d_pack_pair': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, ctypes.c_int64, _c_ptr, _c_ptr, ctypes.c_int64, _c_ptr, _c_ptr]),
    'pasta_conv2d_workspace':       (_c_i64, [ctypes.POINTER(ConvDesc)]),
    'pasta_conv2d_wgrad_workspace': (_c_i64, [ctypes.POINTER(ConvDesc)]),
    'pasta_conv2d_tile':  (ctypes.c_int, [ctypes.POINTER(ConvDesc)]),
    'pasta_conv2d_plan':  (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 5),
    'pasta_conv2d_wgrad_plan': (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_int)]),
    'pasta_conv2d':       (ctypes.c_int, [_c_ptr] * 5 + [ctypes.POINTER(ConvDesc), _c_ptr, _c_i64, _c_ptr]),
    'pasta_conv2d_ex':    (ctypes.c_int, [_c_ptr] * 5 + [ctypes.POINTER(ConvEpilogue), ctypes.POINTER(ConvDesc), _c_ptr, _c_i64, _c_ptr]),
    'pasta_conv2d_wgrad': (ctypes.c_int, [_c_ptr] * 3 + [ctypes.POINTER(ConvDesc), _c_ptr, _c_i64, _c_ptr]),
    'pasta_conv2d_wgrad_modulated': (ctypes.c_int, [_c_ptr] * 6 + [ctypes.POINTER(ConvDesc), _c_ptr, _c_i64, _c_ptr]),
    'pasta_conv2d_wgrad_modulated_workspace': (_c_i64, [ctypes.POINTER(ConvDesc)]),
    'pasta_conv2d_modulated': (ctypes.c_int, [_c_ptr] * 5 + [ctypes.POINTER(ConvEpilogue), ctypes.POINTER(ConvDesc), _c_ptr, _c_i64, _c_ptr]),
    'pasta_tensor_amax':  (ctypes.c_int, [_c_ptr, _c_i64, ctypes.c_int, _c_ptr, _c_ptr]),
    'pasta_demod_coefs':  (ctypes.c_int, [_c_ptr] * 3 + [ctypes.c_int] * 4 + [_c_f32, _c_ptr]),
    'pasta_scale_add':    (ctypes.c_int, [_c_ptr] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_i64, ctypes.c_int, _c_ptr, _c_ptr]),
    'pasta_plane_dot':    (ctypes.c_int, [_c_ptr] * 3 + [ctypes.c_int, _c_i64, _c_i64, _c_ptr]),
    'pasta_mod_bias_act': (ctypes.c_int, [_c_ptr] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_float, ctypes.c_float, _c_ptr, _c_ptr]),
    'pasta_mod_bias_act_bwd_workspace': (_c_i64, [ctypes.c_int, ctypes.c_int, _c_i64]),
    'pasta_mod_bias_act_bwd': (ctypes.c_int, [_c_ptr] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                              ctypes.c_float, ctypes.c_float, _c_ptr, _c_ptr]),
    'pasta_masked_mean_fill': (ctypes.c_int, [_c_ptr] * 5 + [ctypes.c_int, ctypes.c_int, _c_i64, _c_i64, _c_i64, ctypes.c_int, _c_ptr, _c_ptr]),
    'pasta_spade_norm':   (ctypes.c_int, [_c_ptr] * 5 + [ctypes.c_int, _c_i64, _c_i64, _c_f32, ctypes.c_int, _c_f32, _c_f32, ctypes.c_int, _c_i64, _c_ptr, _c_ptr]),
    'pasta_spade_norm_bwd': (ctypes.c_int, [_c_ptr] * 7 + [ctypes.c_int, _c_i64, _c_i64, _c_ptr, ctypes.c_int, _c_f32, _c_f32, ctypes.c_int, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'pasta_ada_matrices': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, ctypes.c_int, ctypes.c_int, _c_ptr, ctypes.POINTER(AdaConfig)] +
                                         [ctypes.c_int] * 4 + [_c_f32, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'pasta_ada_theta':    (ctypes.c_int, [_c_ptr, _c_i64, ctypes.POINTER(_c_f32), ctypes.POINTER(_c_f32), _c_ptr, _c_ptr]),
    'pasta_color_affine': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, ctypes.c_int, _c_ptr]),
    'pasta_ada_grid':     (ctypes.c_int, [_c_ptr, _c_i64, ctypes.c_int, ctypes.c_int, _c_ptr, _c_ptr]),
    'pasta_affine_sample': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64] + [ctypes.c_int] * 5 + [_c_ptr]),
    'pasta_affine_sample_adjoint': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64] + [ctypes.c_int] * 5 + [_c_ptr]),
    'pasta_nan_to_num_multi': (ctypes.c_int, [ctypes.POINTER(_c_ptr), ctypes.POINTER(_c_i64), ctypes.c_int, _c_f32, _c_f32, _c_f32, _c_ptr]),
    'pasta_warp_perspective_u8': (ctypes.c_int, [_c_ptr] * 5 + [ctypes.c_int] * 7 + [_c_ptr]),
    'pasta_patch_composite_u8': (ctypes.c_int, [_c_ptr] * 6 + [ctypes.c_int] * 6 + [_c_ptr]),
}

def get_plugin(module_name='pasta_hip', sources=None, **build_kwargs):
    """Return the loaded library (a ``ctypes.CDLL`` with typed entry points).

    ``sources`` is accepted for signature compatibility with the reference loader and
    ignored: the library always contains every kernel under ``csrc/``."""
    del sources
    with _lock:
        if module_name in _cached_plugins:
            return _cached_plugins[module_name]
        lib_path = build(**build_kwargs)
        # same-box A/B of two builds (tools/ab_lib.sh): PASTA_LIB_AB names another library built from THIS tree with other compile flags
        ab = os.environ.get('PASTA_LIB_AB')
        if ab:
            if not os.path.isfile(ab):
                raise RuntimeError('PASTA_LIB_AB names %r, which does not exist' % ab)
            warnings.warn('PASTA_LIB_AB: loading %s instead of %s (same-box A/B measurement)' % (ab, lib_path))
            lib_path = ab
        lib = ctypes.CDLL(lib_path)
        for name, (restype, argtypes) in ABI.items():
            fn = getattr(lib, name)       # AttributeError here = header/library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        # the structure layouts above (ConvDesc, ConvEpilogue, ...) are those of ONE ABI revision: a library of another one -- a stale A/B
        # build, a copy from another checkout -- would read pointer fields at the wrong offsets
        if lib.pasta_abi_version() != EXPECTED_ABI:
            raise RuntimeError('%s reports ABI %d, this tree expects %d (include/pasta_hip.h): rebuild it from this tree'
                               % (lib_path, lib.pasta_abi_version(), EXPECTED_ABI))
        _cached_plugins[module_name] = lib
        return lib

def check(lib, status):
    """Turn a non-zero status into the RuntimeError the reference's TORCH_CHECK raises."""
    if status != 0:
        raise RuntimeError(lib.pasta_last_error().decode(errors='replace'))

#----------------------------------------------------------------------------
