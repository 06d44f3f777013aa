"""``conv2d`` / ``conv_transpose2d`` on the MI355X implicit-GEMM kernels, with gradients of any order.

Public surface mirrors the reference module torch_utils/ops/conv2d_gradfix.py (``enabled`` :22,
``weight_gradients_disabled`` :23, ``no_weight_gradients`` :25-31, ``conv2d`` :35-38,
``conv_transpose2d`` :40-43). Where the reference forwards to ATen/cuDNN, this module launches
``pasta_conv2d`` / ``pasta_conv2d_wgrad`` (csrc/conv_igemm.hip, fp32 matrix cores). The autograd
structure is the reference's (:107-165): the input gradient is the transposed operator applied
to ``dy``; the weight gradient is its own Function whose backward is again expressed with the
forward operators, so R1's double backward works.
"""

import contextlib
import ctypes

import torch

from . import _native
from .. import custom_ops

#----------------------------------------------------------------------------

enabled = True                      # Kept for API compatibility; the HIP path is always used for GPU tensors.
# Matrix-core arithmetic of the convolutions (include/pasta_hip.h PASTA_MATH_*): 'f16x3' (= 'default': fp32-equivalent
# products from three fp16 MFMAs with power-of-two operand scales), 'bf16x6' (fp32-equivalent from six bf16 MFMAs; the default
# until round 3) or 'f32' (fp32 MFMA, bit-exact fp32 FMA chains); 'bf16x3' / 'bf16' are opt-in reduced modes.  Overridable
# with PASTA_CONV_MATH.
import os as _os
MATH_CODES = {'default': 0, 'f32': 1, 'bf16x6': 2, 'bf16x3': 3, 'bf16': 4, 'f16x3': 5}
conv_math = _os.environ.get('PASTA_CONV_MATH', 'default')
assert conv_math in MATH_CODES, f'PASTA_CONV_MATH must be one of {sorted(MATH_CODES)}'
weight_gradients_disabled = False   # Forcefully disable computation of gradients with respect to the weights.

@contextlib.contextmanager
def no_weight_gradients():
    """Skip weight gradients inside the block (used around R1 / path-length ``autograd.grad``)."""
    global weight_gradients_disabled
    old = weight_gradients_disabled
    weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = old

#----------------------------------------------------------------------------

def _pair(v):
    v = tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    assert len(v) == 2 and all(isinstance(e, int) for e in v)
    return v

class _Cfg(tuple):
    """(transposed, stride, pad_h, pad_w, outpad_h, outpad_w, groups[, wgain]) -- hashable op configuration.
    ``wgain``: the weights enter every launch of the family as ``w * wgain`` (folded into the weight packing, and
    into the weight gradient's reduction), so a layer's ``self.weight * self.weight_gain`` needs no kernel of its own."""
    __slots__ = ()
    wgain = property(lambda s: s[7] if len(s) > 7 else 1.0)
    transposed = property(lambda s: s[0]); stride = property(lambda s: s[1])
    pad_h = property(lambda s: s[2]); pad_w = property(lambda s: s[3])
    outpad_h = property(lambda s: s[4]); outpad_w = property(lambda s: s[5]); groups = property(lambda s: s[6])

def _out_hw(cfg, h, w, kh, kw):
    if cfg.transposed:
        return ((h - 1) * cfg.stride - 2 * cfg.pad_h + kh + cfg.outpad_h,
                (w - 1) * cfg.stride - 2 * cfg.pad_w + kw + cfg.outpad_w)
    return ((h + 2 * cfg.pad_h - kh) // cfg.stride + 1, (w + 2 * cfg.pad_w - kw) // cfg.stride + 1)

IO_CODES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 3}      # pasta_conv_desc.io_dtype (PASTA_F32 / _F16 / _BF16)

_SCOPE = _os.environ.get('PASTA_F16X3_SCOPE', 'all')       # diagnostic: 'fwd' / 'wgrad' = the three-product arithmetic for those launches only

def _desc(cfg, x_shape, c_out, oh, ow, kh, kw, io=torch.float32, kind='fwd'):
    n, c_in, h, w = x_shape
    if _SCOPE != 'all' and _SCOPE != kind and conv_math in ('default', 'f16x3'):
        d = custom_ops.ConvDesc(N=n, C_in=c_in, H=h, W=w, C_out=c_out, OH=oh, OW=ow, kh=kh, kw=kw, stride=cfg.stride,
                                pad_h=cfg.pad_h, pad_w=cfg.pad_w, groups=cfg.groups, transposed=int(cfg.transposed), flip=0,
                                math=MATH_CODES['bf16x6'], wscale=float(cfg.wgain), io_dtype=IO_CODES[io])
        return d
    return custom_ops.ConvDesc(N=n, C_in=c_in, H=h, W=w, C_out=c_out, OH=oh, OW=ow, kh=kh, kw=kw, stride=cfg.stride,
                               pad_h=cfg.pad_h, pad_w=cfg.pad_w, groups=cfg.groups, transposed=int(cfg.transposed), flip=0,
                               math=MATH_CODES[conv_math], wscale=float(cfg.wgain), io_dtype=IO_CODES[io])

# 16-bit storage (fp16 / bf16 tensors in HBM, one matrix-core product per multiply-add, fp32 accumulation; include/pasta_hip.h,
# pasta_conv_desc.io_dtype) exists for the shapes the matrix-core kernel family covers; the planners say which.  Everything
# else -- the few-channel stems and heads -- is converted to fp32 for that launch and back.
_native16_cache = {}

def _native16(kind, desc, has_iscale=False):
    key = (kind, desc.N, desc.C_in, desc.H, desc.W, desc.C_out, desc.OH, desc.OW, desc.kh, desc.kw, desc.stride, desc.pad_h, desc.pad_w,
           desc.groups, desc.transposed, desc.io_dtype, bool(has_iscale))
    hit = _native16_cache.get(key)
    if hit is None:
        lib = _native.lib()
        if kind == 'conv':
            hit = lib.pasta_conv2d_plan(ctypes.byref(desc), int(bool(has_iscale)), None, None, None, None, None) == 0
        else:
            hit = lib.pasta_conv2d_wgrad_plan(ctypes.byref(desc), None) == 0
        _native16_cache[key] = hit
    return hit

# PASTA_MATH_F16X3 (include/pasta_hip.h): the power-of-two operand scales of the ACTIVATIONS come from the tensors' largest
# magnitudes (the weights are scaled per output row by their packing kernel at every launch: nothing about a weight is cached).
# A tensor is scanned by ``pasta_tensor_amax`` (256 partial maxima, one pass at HBM rate) or arrives with the maxima its
# producer kernel left behind (_native.amax_attach), and the result travels with the Python tensor object while its version
# stands: a forward input serves the forward launch and, from the saved tensor, the weight gradient; a gradient serves the
# input-gradient and the weight-gradient launch of the same backward.
#
# What may be cached (round 4): the key (t._version, t.data_ptr()) cannot see a write that goes
# around the version counter (``t.data.mul_()``, numpy / DLPack aliases).  The tensors such idioms are applied to are LEAVES a
# caller holds: parameters (no longer scanned at all) and input batches.  So a scan result is attached only to
#   * tensors with a ``grad_fn`` (results of recorded operations: only autograd-visible code writes them), and
#   * tensors met inside a backward pass (gradients, saved activations being differentiated),
# and a leaf met outside a backward pass -- an input batch, a parameter used as an activation, a no-grad intermediate of a
# torch operator -- is scanned at every use (its maxima are never read from the object: ``tensor_amax`` applies the same predicate
# when it READS a scan result, so a result attached to a leaf inside a backward pass is not found by a later forward).  Producer rows are written by the
# kernel that wrote the tensor and are trusted for that version.  Inference tensors (``torch.inference_mode()``) track no
# version: nothing is cached on them or read back from them.
AMAX_PARTS = 256
_f16x3_cache = {}
_SCAN_TRACE = {} if _os.environ.get('PASTA_AMAX_TRACE') else None
if _SCAN_TRACE is not None:
    import atexit as _atexit
    def _dump_scans():
        import sys
        rows = sorted(_SCAN_TRACE.items(), key=lambda kv: -kv[1] * max(1, int(torch.Size(kv[0][0]).numel())))
        for (shape, src), n in rows[:60]:
            print(f'amax scan x{n:5d}  {str(shape):22s} {src}', file=sys.stderr)
    _atexit.register(_dump_scans)

# PASTA_CHECK_FINITE=1 (debug; synchronises): every convolution under the three-product arithmetic checks its output for
# inf / NaN and raises -- what an operand scale taken from stale maxima (|v S| > 65504) or a non-finite operand produces.
_CHECK_FINITE = _os.environ.get('PASTA_CHECK_FINITE', '0') == '1'

_graph_task_id = getattr(torch._C, '_current_graph_task_id', None)

def _in_backward():
    return _graph_task_id is not None and _graph_task_id() != -1

def _amax_cacheable(t):
    """May a scan of ``t`` be attached to the tensor object (see the note above)?"""
    if t.is_inference():
        return False
    return t.grad_fn is not None or _in_backward()

def tensor_amax(t):
    """[256] partial |max| of a contiguous fp32 GPU tensor; cached on the tensor only where no write can bypass the key."""
    hit = None if t.is_inference() else getattr(t, '_pasta_amax', None)
    # the read side applies the rule of the write side: a SCAN result found on a leaf outside a backward pass is not trusted --
    # it may have been attached while the leaf was met inside one (an image batch whose forward ran a kernel that takes no maxima, scanned by
    # the weight gradient), and a `.data` write since then is invisible to the key.  Producer rows (hit[3]) stand for their version.
    if hit is not None and hit[0] == t._version and hit[1] == t.data_ptr() and (len(hit) > 3 or _amax_cacheable(t)):
        return hit[2]
    if _SCAN_TRACE is not None:                        # diagnostic (PASTA_AMAX_TRACE=1): which tensors still cost a scan
        import sys as _sys
        fr, chain = _sys._getframe(1), []
        while fr is not None and len(chain) < 8:           # the callers inside this package, innermost first
            if 'pasta-gan_amd' in fr.f_code.co_filename and fr.f_code.co_name not in ('apply', '_call_impl', '_wrapped_call_impl'):
                chain.append(f'{fr.f_code.co_name}:{fr.f_lineno}')
            fr = fr.f_back
        key = (tuple(t.shape), (type(t.grad_fn).__name__ if t.grad_fn is not None else ('param' if t.requires_grad else 'plain'))
               + ' <- ' + ' <- '.join(chain) + (' [stale attr]' if hit is not None else ''))
        _SCAN_TRACE[key] = _SCAN_TRACE.get(key, 0) + 1
    parts = torch.empty([AMAX_PARTS], dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        _native.check(_native.lib().pasta_tensor_amax(_native.ptr(t), t.numel(), 0, _native.ptr(parts), _native.stream()))
    if _amax_cacheable(t):
        try:
            t._pasta_amax = (t._version, t.data_ptr(), parts)
        except AttributeError:
            pass
    return parts

def _check_finite(y, what):
    if _CHECK_FINITE and not bool(torch.isfinite(y).all()):
        raise RuntimeError(f'{what}: non-finite output under PASTA_MATH_F16X3 -- a non-finite operand, or an operand scale taken '
                           f'from maxima that no longer describe the tensor (written behind the version counter?)')

def _runs_f16x3(kind, desc, flags=0):
    """Does this launch run the three-product fp16 arithmetic (then it wants the operands' partial maxima)?"""
    if desc.math not in (0, MATH_CODES['f16x3']) or desc.io_dtype != 0:
        return False
    key = (kind, flags & 9, desc.N, desc.C_in, desc.H, desc.W, desc.C_out, desc.OH, desc.OW, desc.kh, desc.kw, desc.stride, desc.pad_h,
           desc.pad_w, desc.groups, desc.transposed)
    hit = _f16x3_cache.get(key)
    if hit is None:
        lib = _native.lib()
        if kind == 'conv':
            math = ctypes.c_int()
            hit = lib.pasta_conv2d_plan(ctypes.byref(desc), int(flags), None, None, ctypes.byref(math), None, None) == 0 and math.value == MATH_CODES['f16x3']
        else:
            kernel = ctypes.c_int()
            hit = lib.pasta_conv2d_wgrad_plan(ctypes.byref(desc), ctypes.byref(kernel)) == 0 and kernel.value in (2, 3, 4)
        _f16x3_cache[key] = hit
    return hit

# Optional measurement hook (bench.py): when set, called as hook(kind, desc, launch) around every native
# convolution launch; ``launch()`` performs it. None = no overhead.
launch_hook = None

def _f32(t):
    return t if t.dtype == torch.float32 else t.float()

def _launch_conv(x, w, cfg, iscale=None, oscale=None, epilogue=None, wmod=None, noise=None, x2=None, used=None, pieces=None, prepacked=None):
    """y = conv(x * iscale[n,c]) * oscale[n,c'] through ``pasta_conv2d_ex``; fp32 accumulation.  fp32 tensors run the
    split-bf16 (fp32-equivalent) or fp32 matrix-core kernels; fp16 / bf16 tensors stay 16-bit in HBM where a kernel exists
    (``_native16``) and are converted for the launch otherwise.
    ``epilogue`` = (bias or None, act code 1..3, alpha, gain, clamp[, residual or None]) fuses Conv2dLayer's bias_act into the
    store; the residual ([N, C_out, OH, OW]) is added to the convolution before the bias.
    ``noise`` = (plane(s) [OH, OW] or [N, 1, OH, OW] fp32, strength scalar tensor): added after ``oscale`` (needs ``epilogue``).
    ``wmod`` = (styles [G, I], dcoefs [G, O] or None): ``w`` is ONE group's weight shared by all ``cfg.groups`` groups and
    modulated per group by the packing kernel (``pasta_conv2d_modulated``).
    ``x2`` ([N, C2, H, W], pointwise convolutions only -- ``cat1x1_available``): the convolution runs over the channel concatenation
    ``cat([x, x2], 1)`` without forming it (``pasta_conv_desc.x2``).
    ``used`` (dict, optional): receives ``'x_amax'`` / ``'x2_amax'`` = the partial maxima the launch took for its activations, so that the
    autograd node can hand them to the weight gradient of the SAME saved tensor (a saved non-leaf tensor comes back from autograd as a new
    Python object without the attribute the maxima travel on: it would be scanned a second time).
    ``pieces`` = (bound row [256], logical shape (N, C, H, W)): ``x`` is not an NCHW tensor but the producer-written operand of the three-product
    arithmetic (``blur_pieces``: PASTA_LAYOUT_PIECES16, include/pasta_hip.h) -- the launch copies its sixteen-byte pieces instead of splitting.
    ``prepacked`` = the workspace tensor of THIS launch with the weights already packed in it (``pack_pair``): no packing kernel."""
    _native.require_gpu(x, 'conv2d')
    if pieces is not None:
        assert iscale is None and oscale is None and wmod is None and noise is None and x2 is None
        return _launch_conv_pieces(x, w, cfg, epilogue, pieces)
    if x.ndim != 4 or w.ndim != 4:
        raise RuntimeError('conv2d: x and w must be rank 4')
    out_dtype = x.dtype
    w = _f32(w).contiguous()
    kh, kw = w.shape[2], w.shape[3]
    shared = cfg.groups if wmod is not None else 1        # the weight tensor holds one group
    c_in = x.shape[1] + (x2.shape[1] if x2 is not None else 0)
    if cfg.transposed:
        if w.shape[0] * shared != c_in:
            raise RuntimeError(f'conv_transpose2d: weight {tuple(w.shape)} does not match input channels {c_in}')
        c_out = w.shape[1] * cfg.groups
    else:
        if w.shape[1] * cfg.groups != c_in:
            raise RuntimeError(f'conv2d: weight {tuple(w.shape)} does not match input channels {c_in} (groups={cfg.groups})')
        c_out = w.shape[0] * shared
    oh, ow = _out_hw(cfg, x.shape[2], x.shape[3], kh, kw)
    if oh < 1 or ow < 1:
        raise RuntimeError('conv2d: output must be at least 1x1')
    if x.numel() == 0 or c_out * oh * ow == 0:
        return torch.zeros([x.shape[0], c_out, oh, ow], dtype=out_dtype, device=x.device)
    io = torch.float32
    if x.dtype in (torch.float16, torch.bfloat16) and _native16('conv', _desc(cfg, x.shape, c_out, oh, ow, kh, kw, x.dtype), iscale is not None):
        io = x.dtype
    x = x.contiguous() if io is not torch.float32 else _f32(x).contiguous()
    y = torch.empty([x.shape[0], c_out, oh, ow], dtype=io, device=x.device)
    desc = _desc(cfg, (x.shape[0], c_in, x.shape[2], x.shape[3]), c_out, oh, ow, kh, kw, io)
    if x2 is not None:
        if io is not torch.float32 or x2.dtype != torch.float32 or x2.shape[0] != x.shape[0] or x2.shape[2:] != x.shape[2:]:
            raise RuntimeError('conv2d: the second input tensor must be fp32 and match the first in batch and plane size')
        x2 = x2.contiguous()
        desc.x2, desc.C1 = x2.data_ptr(), int(x.shape[1])
    lib = _native.lib()
    nbytes = lib.pasta_conv2d_workspace(ctypes.byref(desc))
    if nbytes < 0:
        _native.check(1)
    if prepacked is not None and prepacked.numel() * 4 >= nbytes and iscale is None and oscale is None and wmod is None and x2 is None and io is torch.float32:
        work = prepacked
        desc.w_prepacked = 1
    else:
        work = torch.empty([max(nbytes // 4, 4)], dtype=torch.float32, device=x.device)
    if iscale is not None:
        iscale = _f32(iscale).contiguous()
        assert iscale.shape == (x.shape[0], x.shape[1])
    if oscale is not None:
        oscale = _f32(oscale).contiguous()
        assert oscale.shape == (x.shape[0], c_out)
    ep = y_row = None
    if epilogue is not None:
        bias, act_code, alpha, gain, clamp = epilogue[:5]
        res = epilogue[5] if len(epilogue) > 5 else None
        bias = _f32(bias).contiguous() if bias is not None else None
        if res is not None:
            res = res.to(io).contiguous()
            if res.shape != y.shape:
                raise RuntimeError(f'conv2d: residual {tuple(res.shape)} does not match the output {tuple(y.shape)}')
        nz = nstr = None
        if noise is not None:
            nz, nstr = _f32(noise[0]).contiguous(), _f32(noise[1]).reshape(1).contiguous()
            if nz.numel() not in (oh * ow, x.shape[0] * oh * ow):
                raise RuntimeError(f'conv2d: noise {tuple(nz.shape)} is neither one {oh}x{ow} plane nor one per sample')
        ep = custom_ops.ConvEpilogue(bias=bias.data_ptr() if bias is not None else None, act=int(act_code), alpha=float(alpha),
                                     gain=float(gain), clamp=float(clamp), res=res.data_ptr() if res is not None else None,
                                     noise=nz.data_ptr() if nz is not None else None, noise_strength=nstr.data_ptr() if nz is not None else None,
                                     noise_per_sample=int(nz is not None and nz.numel() != oh * ow))
        y_row = _native.amax_slot(y)            # the layer's output usually feeds the next convolution
        if y_row is not None:
            ep.y_amax = y_row.data_ptr()
    elif noise is not None:
        raise RuntimeError('conv2d: the noise operand is part of the fused epilogue')
    if wmod is not None:
        assert iscale is None and oscale is None
        mod_s = _f32(wmod[0]).contiguous()
        mod_d = _f32(wmod[1]).contiguous() if wmod[1] is not None else None
        assert mod_s.numel() == x.shape[1] and (mod_d is None or mod_d.numel() == c_out)
    x_amax = None
    f16x3 = _runs_f16x3('conv', desc, (1 if iscale is not None else 0) | (8 if wmod is not None else 0))
    x2_amax = None
    if f16x3:
        x_amax = tensor_amax(x)                 # the weights are scaled per output row by their packing kernel: nothing to pass
        desc.x_amax = x_amax.data_ptr()
        if x2 is not None:
            x2_amax = tensor_amax(x2)
            desc.x2_amax = x2_amax.data_ptr()
        if used is not None:
            used['x_amax'], used['x2_amax'] = x_amax, x2_amax
    def launch():
        with torch.cuda.device(x.device):
            if wmod is not None:
                st = lib.pasta_conv2d_modulated(_native.ptr(x), _native.ptr(w), _native.ptr(mod_s), _native.ptr(mod_d), _native.ptr(y),
                                                ctypes.byref(ep) if ep is not None else None, ctypes.byref(desc), _native.ptr(work),
                                                work.numel() * 4, _native.stream())
            else:
                st = lib.pasta_conv2d_ex(_native.ptr(x), _native.ptr(w), _native.ptr(y), _native.ptr(iscale), _native.ptr(oscale),
                                         ctypes.byref(ep) if ep is not None else None, ctypes.byref(desc), _native.ptr(work),
                                         work.numel() * 4, _native.stream())
        _native.check(st)
    if launch_hook is None:
        launch()
    else:
        # 'conv_isc': the launch passes an input scale (another kernel instance); flags = PASTA_PLAN_* of include/pasta_hip.h
        flags = (1 if iscale is not None else 0) | (2 if oscale is not None else 0) | (4 if ep is not None else 0) | (8 if wmod is not None else 0)
        launch_hook('conv' if iscale is None else 'conv_isc', desc, launch, flags)
    if f16x3:
        _check_finite(y, 'conv2d')
    if y_row is not None and y.dtype == out_dtype:
        _native.amax_attach(y, y_row)
    return y.to(out_dtype)

# ---- producer-written operand pieces (round 5; csrc/pieces.hip, include/pasta_hip.h "Producer-written operand pieces") --------------------
# The low-pass in front of a stride-2 convolution (reference conv2d_resample.py:119-122) has ONE reader besides that convolution's weight
# gradient, and its magnitude is bounded by its input's: it can write its output once as the fp16 pieces h | l' of the three-product
# arithmetic (16-byte units of eight channels, [N][C / 8][H][piece][W]), and both consumers copy pieces instead of fetching fp32 with eight
# channel-strided loads per unit and splitting it -- in every launch that touches the tensor.
_PIECES = _os.environ.get('PASTA_PIECES', '1') != '0'       # A/B switch: 0 = fp32 blurred tensor, as before
_pieces_cache = {}

def pieces_available(x, f, weight, pad4, groups=1):
    """Can ``conv2d(upfirdn2d(x, f, padding=pad4), weight, stride=2)`` run on the blurred tensor as producer-written pieces (forward kernel 10 and
    weight-gradient kernel 6 of the planners)?  fp32 tensors, the default arithmetic, a 4 x 4 filter, whole channel octets."""
    if not (_PIECES and x.device.type == 'cuda' and x.dtype == torch.float32 and weight.dtype == torch.float32 and conv_math in ('default', 'f16x3')
            and groups == 1 and f is not None and f.ndim == 2 and tuple(f.shape) == (4, 4) and x.ndim == 4 and x.shape[1] % 8 == 0
            and tuple(weight.shape[2:]) == (3, 3) and weight.shape[1] == x.shape[1] and x.numel() > 0 and _SCOPE == 'all'):
        return False
    px0, px1, py0, py1 = pad4
    n, c, h, wd = (int(v) for v in x.shape)
    bh, bw = h + py0 + py1 - 3, wd + px0 + px1 - 3
    key = (n, c, bh, bw, int(weight.shape[0]))
    hit = _pieces_cache.get(key)
    if hit is None:
        hit = False
        if bh >= 3 and bw >= 3:
            cfg = _Cfg((False, 2, 0, 0, 0, 0, 1, 1.0))
            oh, ow = _out_hw(cfg, bh, bw, 3, 3)
            desc = _desc(cfg, (n, c, bh, bw), int(weight.shape[0]), oh, ow, 3, 3)
            desc.x_layout = 1
            lib, k = _native.lib(), ctypes.c_int()
            hit = lib.pasta_conv2d_plan(ctypes.byref(desc), 4, None, None, None, None, ctypes.byref(k)) == 0 and k.value == 10
            hit = hit and lib.pasta_conv2d_wgrad_plan(ctypes.byref(desc), ctypes.byref(k)) == 0 and k.value == 6
        _pieces_cache[key] = hit
    return hit

def blur_pieces(x, f, pad4, flip_filter=False, gain=1.0, x_amax=None):
    """``upfirdn2d(x, f, padding=pad4, flip_filter, gain)`` for a 4 x 4 filter, written as PASTA_LAYOUT_PIECES16.  Returns (pieces -- a uint8
    tensor of N * C / 8 * OH * OW * 32 bytes --, bound row [256] fp32, logical shape (N, C, OH, OW)); the bound row is ``x_amax`` (the partial
    maxima of x: its producer's row, or one scan) times gain * sum |f| and fixes the operand's power-of-two scale on both sides."""
    _native.require_gpu(x, 'blur_pieces')
    x = x.contiguous()
    n, c, h, wd = (int(v) for v in x.shape)
    px0, px1, py0, py1 = pad4
    oh, ow = h + py0 + py1 - 3, wd + px0 + px1 - 3
    lib = _native.lib()
    nbytes = lib.pasta_pieces_bytes(n, c, oh, ow)
    if nbytes < 0 or oh < 1 or ow < 1:
        raise RuntimeError(f'blur_pieces: {tuple(x.shape)} -> {oh} x {ow}: the channel count must be a multiple of 8 and the output at least 1 x 1')
    parts = tensor_amax(x) if x_amax is None else x_amax
    pieces = torch.empty([nbytes], dtype=torch.uint8, device=x.device)
    bound = torch.empty([AMAX_PARTS], dtype=torch.float32, device=x.device)
    f = f.contiguous()
    def launch():
        with torch.cuda.device(x.device):
            _native.check(lib.pasta_blur_pieces(_native.ptr(x), _native.ptr(f), _native.ptr(pieces), _native.ptr(parts), _native.ptr(bound), n, c, h, wd,
                                                int(px0), int(px1), int(py0), int(py1), int(bool(flip_filter)), float(gain), _native.stream()))
    from . import upfirdn2d as _up
    if _up.launch_hook is None:
        launch()
    else:       # algorithmic bytes: fp32 in, 4 bytes per element out (two fp16 pieces)
        _up.launch_hook((x.numel() + n * c * oh * ow) * 4, (tuple(x.shape), 1, 1, 4, 'pieces'), launch)
    return pieces, bound, (n, c, oh, ow)

def pieces_pack(x, x_amax=None):
    """An fp32 NCHW tensor as PASTA_LAYOUT_PIECES16, split exactly as the consuming kernels split it in their staging (``pasta_pieces_pack``).
    Returns (pieces, bound row, shape) like ``blur_pieces``; the bound row is x's own partial maxima."""
    _native.require_gpu(x, 'pieces_pack')
    x = x.float().contiguous()
    n, c, h, wd = (int(v) for v in x.shape)
    lib = _native.lib()
    nbytes = lib.pasta_pieces_bytes(n, c, h, wd)
    if nbytes < 0:
        raise RuntimeError(f'pieces_pack: {tuple(x.shape)}: the channel count must be a multiple of 8')
    parts = tensor_amax(x) if x_amax is None else x_amax
    pieces = torch.empty([nbytes], dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _native.check(lib.pasta_pieces_pack(_native.ptr(x), _native.ptr(parts), _native.ptr(pieces), n, c, h, wd, _native.stream()))
    return pieces, parts, (n, c, h, wd)

# ---- one operand, several 3x3 layers (round 5): a tensor that several stride-1 3x3 convolutions read (the SPADE feature map: three
# conv_mlp batches per generator pass, networks.py Spade_ResBlockV2) is packed ONCE and every forward launch copies the pieces (plan kernel 7
# with x_layout = PASTA_LAYOUT_PIECES16: +10 % on 256 -> 384 at 128 x 128, profiles/r5_s1_pieces_microbench.txt).  The pieces ride on the tensor
# object, keyed by version and address like the maxima rows; the weight gradients keep reading the fp32 tensor.
_SHARED_PIECES = _os.environ.get('PASTA_SHARED_PIECES', '1') != '0'       # A/B switch: 0 = every launch splits the fp32 tensor itself
_shared_cache = {}

def _shared_pieces_ok(x, weight, cfg, has_epilogue):
    """Does the eight-wave stride-1 tile kernel take conv2d(x, weight) with x as pieces?"""
    if not (_SHARED_PIECES and _PIECES and x.device.type == 'cuda' and x.dtype == torch.float32 and weight.dtype == torch.float32
            and conv_math in ('default', 'f16x3') and _SCOPE == 'all' and cfg.groups == 1 and cfg.stride == 1 and not cfg.transposed
            and tuple(weight.shape[2:]) == (3, 3) and x.ndim == 4 and x.shape[1] % 8 == 0):
        return False
    n, c, h, wd = (int(v) for v in x.shape)
    key = (n, c, h, wd, int(weight.shape[0]), cfg.pad_h, cfg.pad_w, bool(has_epilogue))
    hit = _shared_cache.get(key)
    if hit is None:
        oh, ow = _out_hw(cfg, h, wd, 3, 3)
        desc = _desc(cfg, (n, c, h, wd), int(weight.shape[0]), oh, ow, 3, 3)
        desc.x_layout = 1
        k = ctypes.c_int()
        hit = _native.lib().pasta_conv2d_plan(ctypes.byref(desc), 4 if has_epilogue else 0, None, None, None, None, ctypes.byref(k)) == 0 and k.value == 7
        _shared_cache[key] = hit
    return hit

def share_pieces(x):
    """Pack ``x`` for the stride-1 3x3 layers that will read it (a no-op where none of them could use the pieces); returns ``x``."""
    if (_SHARED_PIECES and _PIECES and x.device.type == 'cuda' and x.dtype == torch.float32 and x.ndim == 4 and x.shape[1] % 8 == 0 and x.shape[1] >= 16
            and conv_math in ('default', 'f16x3') and _SCOPE == 'all' and not x.is_inference() and x.is_contiguous()
            and not torch.cuda.is_current_stream_capturing() and _shared_lookup(x) is None):
        pieces, bound, shape = pieces_pack(x)
        try:
            x._pasta_pieces = (x._version, x.data_ptr(), pieces, bound, shape)
        except AttributeError:
            pass
    return x

def _shared_lookup(x):
    hit = getattr(x, '_pasta_pieces', None)
    if hit is not None and hit[0] == x._version and hit[1] == x.data_ptr():
        return hit
    return None

def pieces_unpack(pieces, bound, shape):
    """(h + 2^-11 l') / S as an fp32 NCHW tensor (tests, diagnostics): the 22 bits the consumers multiply."""
    n, c, h, wd = shape
    y = torch.empty([n, c, h, wd], dtype=torch.float32, device=pieces.device)
    with torch.cuda.device(pieces.device):
        _native.check(_native.lib().pasta_pieces_unpack(_native.ptr(pieces), _native.ptr(bound), _native.ptr(y), n, c, h, wd, _native.stream()))
    return y

def _launch_conv_pieces(x, w, cfg, epilogue, pieces):
    """The stride-2 forward convolution on a producer-written operand (``_launch_conv`` with ``pieces``)."""
    bound, (n, c_in, h, wd) = pieces
    w = _f32(w).contiguous()
    kh, kw = w.shape[2], w.shape[3]
    assert cfg.groups == 1 and w.shape[0 if cfg.transposed else 1] == c_in
    c_out = w.shape[1 if cfg.transposed else 0]
    oh, ow = _out_hw(cfg, h, wd, kh, kw)
    y = torch.empty([n, c_out, oh, ow], dtype=torch.float32, device=x.device)
    desc = _desc(cfg, (n, c_in, h, wd), c_out, oh, ow, kh, kw)
    desc.x_layout, desc.x_amax = 1, bound.data_ptr()
    lib = _native.lib()
    nbytes = lib.pasta_conv2d_workspace(ctypes.byref(desc))
    if nbytes < 0:
        _native.check(1)
    work = torch.empty([max(nbytes // 4, 4)], dtype=torch.float32, device=x.device)
    ep = y_row = None
    if epilogue is not None:
        bias, act_code, alpha, gain, clamp = epilogue[:5]
        res = epilogue[5] if len(epilogue) > 5 else None
        bias = _f32(bias).contiguous() if bias is not None else None
        if res is not None:
            res = res.float().contiguous()
            if res.shape != y.shape:
                raise RuntimeError(f'conv2d: residual {tuple(res.shape)} does not match the output {tuple(y.shape)}')
        ep = custom_ops.ConvEpilogue(bias=bias.data_ptr() if bias is not None else None, act=int(act_code), alpha=float(alpha), gain=float(gain),
                                     clamp=float(clamp), res=res.data_ptr() if res is not None else None, noise=None, noise_strength=None, noise_per_sample=0)
        y_row = _native.amax_slot(y)
        if y_row is not None:
            ep.y_amax = y_row.data_ptr()
    def launch():
        with torch.cuda.device(x.device):
            st = lib.pasta_conv2d_ex(_native.ptr(x), _native.ptr(w), _native.ptr(y), None, None, ctypes.byref(ep) if ep is not None else None,
                                     ctypes.byref(desc), _native.ptr(work), work.numel() * 4, _native.stream())
        _native.check(st)
    if launch_hook is None:
        launch()
    else:
        launch_hook('conv', desc, launch, 4 if ep is not None else 0)
    _check_finite(y, 'conv2d')
    if y_row is not None:
        _native.amax_attach(y, y_row)
    return y

def _launch_wgrad_pieces(pieces_x, dy, cfg, w_shape, pieces, out_dtype=None):
    """dw of the stride-2 convolution whose input is a producer-written operand (pasta_conv2d_wgrad_plan kernel 6)."""
    bound, (n, c_in, h, wd) = pieces
    out_dtype = dy.dtype if out_dtype is None else out_dtype
    dy = _f32(dy).contiguous()
    dw = torch.empty(list(w_shape), dtype=torch.float32, device=dy.device)
    desc = _desc(cfg, (n, c_in, h, wd), dy.shape[1], dy.shape[2], dy.shape[3], w_shape[2], w_shape[3], kind='wgrad')
    desc.x_layout, desc.x_amax = 1, bound.data_ptr()
    amax_dy = tensor_amax(dy)
    desc.dy_amax = amax_dy.data_ptr()
    lib = _native.lib()
    nbytes = lib.pasta_conv2d_wgrad_workspace(ctypes.byref(desc))
    if nbytes < 0:
        _native.check(1)
    work = torch.empty([max(nbytes // 4, 4)], dtype=torch.float32, device=dy.device)
    def launch():
        with torch.cuda.device(dy.device):
            st = lib.pasta_conv2d_wgrad(_native.ptr(pieces_x), _native.ptr(dy), _native.ptr(dw), ctypes.byref(desc), _native.ptr(work), work.numel() * 4, _native.stream())
        _native.check(st)
    if launch_hook is None:
        launch()
    else:
        launch_hook('wgrad', desc, launch, 0)
    _check_finite(dw, 'conv2d_wgrad')
    return dw.to(out_dtype)

def _launch_wgrad(x, dy, cfg, w_shape, out_dtype=None, x_amax=None):
    """dw of ``conv(x, w)`` given dy through ``pasta_conv2d_wgrad``; the gradient comes out fp32 (weights are fp32 masters)
    and is returned as ``out_dtype`` (default: dy's)."""
    _native.require_gpu(x, 'conv2d_wgrad')
    out_dtype = dy.dtype if out_dtype is None else out_dtype
    kh, kw = w_shape[2], w_shape[3]
    dw = torch.empty(list(w_shape), dtype=torch.float32, device=x.device)
    if dw.numel() == 0:
        return dw.to(out_dtype)
    if x.numel() == 0 or dy.numel() == 0:
        return dw.zero_().to(out_dtype)
    io = torch.float32
    if x.dtype == dy.dtype and x.dtype in (torch.float16, torch.bfloat16) and \
            _native16('wgrad', _desc(cfg, x.shape, dy.shape[1], dy.shape[2], dy.shape[3], kh, kw, x.dtype)):
        io = x.dtype
    x = x.contiguous() if io is not torch.float32 else _f32(x).contiguous()
    dy = dy.contiguous() if io is not torch.float32 else _f32(dy).contiguous()
    desc = _desc(cfg, x.shape, dy.shape[1], dy.shape[2], dy.shape[3], kh, kw, io, kind='wgrad')
    lib = _native.lib()
    nbytes = lib.pasta_conv2d_wgrad_workspace(ctypes.byref(desc))
    if nbytes < 0:
        _native.check(1)
    work = torch.empty([max(nbytes // 4, 4)], dtype=torch.float32, device=x.device)
    f16x3 = _runs_f16x3('wgrad', desc)
    if f16x3:
        # x_amax: the maxima the forward launch took for this very tensor (autograd has checked that it was not written since)
        amax_x, amax_dy = (x_amax if x_amax is not None and x.dtype == torch.float32 else tensor_amax(x)), tensor_amax(dy)
        desc.x_amax, desc.dy_amax = amax_x.data_ptr(), amax_dy.data_ptr()
    def launch():
        with torch.cuda.device(x.device):
            st = lib.pasta_conv2d_wgrad(_native.ptr(x), _native.ptr(dy), _native.ptr(dw), ctypes.byref(desc),
                                        _native.ptr(work), work.numel() * 4, _native.stream())
        _native.check(st)
    if launch_hook is None:
        launch()
    else:
        launch_hook('wgrad', desc, launch, 0)
    if f16x3:
        _check_finite(dw, 'conv2d_wgrad')
    return dw.to(out_dtype)

#----------------------------------------------------------------------------

def _grad_cfg(cfg, x_hw, y_hw, kh, kw):
    """Configuration of the operator that maps dy back to dx (reference :95-104, :125-128)."""
    if cfg.transposed:
        return _Cfg((False, cfg.stride, cfg.pad_h, cfg.pad_w, 0, 0, cfg.groups, cfg.wgain))
    oph = x_hw[0] - ((y_hw[0] - 1) * cfg.stride - 2 * cfg.pad_h + kh)
    opw = x_hw[1] - ((y_hw[1] - 1) * cfg.stride - 2 * cfg.pad_w + kw)
    return _Cfg((True, cfg.stride, cfg.pad_h, cfg.pad_w, oph, opw, cfg.groups, cfg.wgain))

# ---- one packing launch for a convolution and its input gradient (round 5; include/pasta_hip.h, pasta_conv2d_pack_pair) -------------------
# Every forward convolution whose input needs a gradient is followed, in the backward pass, by the same weights packed the other way round
# (transposed and mirrored): 120 of the 310 packing launches of a training step.  The forward packs BOTH orientations with one kernel and the
# autograd node carries the second workspace to its backward.
_PACK_PAIR = _os.environ.get('PASTA_PACK_PAIR', '1') != '0'       # A/B switch: 0 = every launch packs for itself

def _pack_pair(x, w, cfg):
    """-> (workspace of the forward launch, workspace of its input-gradient launch), both with the weights packed, or (None, None)."""
    if not (_PACK_PAIR and x.device.type == 'cuda' and x.dtype == torch.float32 and w.dtype == torch.float32 and conv_math in ('default', 'f16x3')
            and _SCOPE == 'all' and x.ndim == 4 and w.ndim == 4 and x.numel() > 0 and not torch.cuda.is_current_stream_capturing()):
        return None, None
    kh, kw = int(w.shape[2]), int(w.shape[3])
    n, c_in, h, wd = (int(v) for v in x.shape)
    c_out = int(w.shape[1]) * cfg.groups if cfg.transposed else int(w.shape[0])
    oh, ow = _out_hw(cfg, h, wd, kh, kw)
    if oh < 1 or ow < 1:
        return None, None
    gcfg = _grad_cfg(cfg, (h, wd), (oh, ow), kh, kw)
    da = _desc(cfg, (n, c_in, h, wd), c_out, oh, ow, kh, kw)
    db = _desc(gcfg, (n, c_out, oh, ow), c_in, h, wd, kh, kw)
    lib = _native.lib()
    na, nb = lib.pasta_conv2d_workspace(ctypes.byref(da)), lib.pasta_conv2d_workspace(ctypes.byref(db))
    if na < 0 or nb < 0:
        return None, None
    wa = torch.empty([max(na // 4, 4)], dtype=torch.float32, device=x.device)
    wb = torch.empty([max(nb // 4, 4)], dtype=torch.float32, device=x.device)
    mask = ctypes.c_int(0)
    wc = w.contiguous()
    with torch.cuda.device(x.device):
        _native.check(lib.pasta_conv2d_pack_pair(_native.ptr(wc), ctypes.byref(da), _native.ptr(wa), wa.numel() * 4, ctypes.byref(db), _native.ptr(wb),
                                                 wb.numel() * 4, _native.stream(), ctypes.byref(mask)))
    return (wa, wb) if mask.value == 3 else (None, None)

class _ConvHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, cfg, prepacked=None):
        used = {}
        ctx.dgrad_ws = None
        if prepacked is None and ctx.needs_input_grad[0]:
            prepacked, ctx.dgrad_ws = _pack_pair(x, w, cfg)
        y = _launch_conv(x, w, cfg, used=used, prepacked=prepacked)
        ctx.save_for_backward(x, w)
        ctx.cfg, ctx.x_amax = cfg, (used.get('x_amax') if x.dtype == torch.float32 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        cfg = ctx.cfg
        dx = dw = None
        if ctx.needs_input_grad[0]:
            gcfg = _grad_cfg(cfg, x.shape[2:], dy.shape[2:], w.shape[2], w.shape[3])
            dx = _ConvHip.apply(dy, w, gcfg, ctx.dgrad_ws)
            assert dx.shape == x.shape
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            dw = _ConvWgradHip.apply(dy, x, cfg, tuple(w.shape), w.dtype, ctx.x_amax)
        return dx, dw, None, None

class _ConvBiasActHip(torch.autograd.Function):
    """y = bias_act(conv(x, w), b) with the bias / activation / gain / clamp applied in the convolution's epilogue.
    The backward is assembled from the stand-alone differentiable pieces (bias_act gradient kernel, input- and
    weight-gradient convolutions), so gradients of any order keep working.

    ``passthrough=True`` returns ``(y, x)``: the second output is the input again, to be handed to the OTHER consumers of ``x`` (a
    residual block's skip branch).  Their gradient then arrives here as ``dxp`` and joins this layer's input gradient in the epilogue of
    the input-gradient launch (one read of it) -- instead of autograd adding two full tensors afterwards (two reads and a write)."""
    @staticmethod
    def forward(ctx, x, w, b, cfg, act_cfg, res=None, passthrough=False, prepacked=None):
        act, alpha, gain, clamp = act_cfg
        from . import bias_act as ba
        used = {}
        ctx.dgrad_ws = None
        shared = _shared_lookup(x) if not x.is_inference() else None
        if shared is not None and _shared_pieces_ok(x, w, cfg, True):
            # x was packed for its several readers (share_pieces): this launch copies the pieces; its maxima are the pack's bound row
            y = _launch_conv(shared[2], w, cfg, epilogue=(b, ba.activation_funcs[act].cuda_idx, alpha, gain, clamp, res), pieces=(shared[3], shared[4]))
            used['x_amax'] = shared[3]
        else:
            if prepacked is None and ctx.needs_input_grad[0]:
                prepacked, ctx.dgrad_ws = _pack_pair(x, w, cfg)
            y = _launch_conv(x, w, cfg, epilogue=(b, ba.activation_funcs[act].cuda_idx, alpha, gain, clamp, res), used=used, prepacked=prepacked)
        if ba.slope_tape is not None:               # test instrument (bias_act.SlopeTape)
            y = ba.slope_tape.visit(y, act)
        # y is needed by the backward only as the activation / clamp mask; a linear, unclamped layer (the residual
        # skips, whose output the blocks then update in place) must not pin it
        keep_y = act != 'linear' or clamp >= 0
        ctx.save_for_backward(x, w, b, y if keep_y else None)
        ctx.cfg, ctx.act_cfg, ctx.x_amax = cfg, act_cfg, (used.get('x_amax') if x.dtype == torch.float32 else None)
        if passthrough:
            ctx.set_materialize_grads(False)        # an unused output's gradient arrives as None, not as a tensor of zeros
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, dxp=None):
        from . import bias_act as ba
        x, w, b, y = ctx.saved_tensors
        if dy is None:                              # only the pass-through output was differentiated
            return dxp, None, None, None, None, None, None, None
        act, alpha, gain, clamp = ctx.act_cfg
        cfg = ctx.cfg
        dz = dy
        dx = dw = db = None
        want_db = b is not None and ctx.needs_input_grad[2]
        if act != 'linear' or gain != 1 or clamp >= 0:
            if want_db:     # derivative of the epilogue and the bias gradient in one pass
                dz, db = ba.grad_with_bias_grad(dy.contiguous(), y, (1, act, alpha, gain, clamp))
            else:
                dz = ba._BiasActHipGrad.apply(dy.contiguous(), None, None, y, (1, act, alpha, gain, clamp))
        if ctx.needs_input_grad[0]:
            gcfg = _grad_cfg(cfg, x.shape[2:], dz.shape[2:], w.shape[2], w.shape[3])
            if dxp is not None and dxp.dtype == dz.dtype:
                dx = _ConvBiasActHip.apply(dz, w, None, gcfg, _LINEAR_EPILOGUE, dxp, False, ctx.dgrad_ws)
            else:
                dx = _ConvHip.apply(dz, w, gcfg, ctx.dgrad_ws)
                if dxp is not None:
                    dx = dx + dxp
            assert dx.shape == x.shape
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            dw = _ConvWgradHip.apply(dz, x, cfg, tuple(w.shape), w.dtype, ctx.x_amax)
        if want_db and db is None:
            db = ba._BiasSum.apply(dz, 1)
        dres = dz if ctx.needs_input_grad[5] else None     # the residual enters before the activation
        return dx, dw, db, None, None, dres, None, None

class _BlurConvS2Hip(torch.autograd.Function):
    """``bias_act(conv2d(upfirdn2d(x, f, padding), w, stride=2) [+ res], b)`` -- the down path of conv2d_resample (reference :119-122 followed by the
    layer's bias_act) -- with the blurred tensor written once as the operand pieces of the default arithmetic (``blur_pieces``) instead of as an
    fp32 tensor: the forward convolution and, in the backward, the weight gradient copy pieces.  The backward is assembled from the stand-alone
    differentiable operators (conv_transpose2d for the input gradient, the mirrored filter behind it), so gradients of any order keep working;
    where a gradient OF the weight gradient may be asked for (a backward pass that records a graph: R1), the weight gradient is formed from the
    fp32 blurred tensor recomputed from ``x`` by the differentiable operators instead.  ``passthrough`` as in ``_ConvBiasActHip``: the filter is
    what reads ``x``, and the other consumers' gradient is the addend of its backward launch."""
    @staticmethod
    def forward(ctx, x, f, w, b, ucfg, cfg, act_cfg, res=None, passthrough=False):
        act, alpha, gain, clamp = act_cfg
        from . import bias_act as ba
        px0, px1, py0, py1, flip, ugain = ucfg
        pieces, bound, shape = blur_pieces(x, f, (px0, px1, py0, py1), flip, ugain)
        y = _launch_conv(pieces, w, cfg, epilogue=(b, ba.activation_funcs[act].cuda_idx, alpha, gain, clamp, res), pieces=(bound, shape))
        if ba.slope_tape is not None:               # test instrument (bias_act.SlopeTape)
            y = ba.slope_tape.visit(y, act)
        keep_y = act != 'linear' or clamp >= 0
        ctx.save_for_backward(x, f, w, b, y if keep_y else None, pieces, bound)
        ctx.ucfg, ctx.cfg, ctx.act_cfg, ctx.shape = ucfg, cfg, act_cfg, shape
        if passthrough:
            ctx.set_materialize_grads(False)
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, dxp=None):
        from . import bias_act as ba
        from . import upfirdn2d as up
        x, f, w, b, y, pieces, bound = ctx.saved_tensors
        if dy is None:
            return dxp, None, None, None, None, None, None, None, None
        act, alpha, gain, clamp = ctx.act_cfg
        cfg, shape = ctx.cfg, ctx.shape
        px0, px1, py0, py1, flip, ugain = ctx.ucfg
        dz = dy
        dx = dw = db = None
        want_db = b is not None and ctx.needs_input_grad[3]
        if act != 'linear' or gain != 1 or clamp >= 0:
            if want_db:
                dz, db = ba.grad_with_bias_grad(dy.contiguous(), y, (1, act, alpha, gain, clamp))
            else:
                dz = ba._BiasActHipGrad.apply(dy.contiguous(), None, None, y, (1, act, alpha, gain, clamp))
        if ctx.needs_input_grad[0]:
            gcfg = _grad_cfg(cfg, shape[2:], dz.shape[2:], w.shape[2], w.shape[3])
            dxb = _ConvHip.apply(dz, w, gcfg)                   # gradient of the blurred tensor
            assert tuple(dxb.shape) == tuple(shape)
            ih, iw = x.shape[2], x.shape[3]
            ucfg_t = (1, 1, 1, 1, 3 - px0, iw - shape[3] + px0, 3 - py0, ih - shape[2] + py0, not flip, ugain)      # upfirdn2d._Upfirdn2dHip.backward, up = down = 1, 4 taps
            dx = up._Upfirdn2dHip.apply(dxb, f, ucfg_t, dxp) if dxp is not None else up._Upfirdn2dHip.apply(dxb, f, ucfg_t)
            assert dx.shape == x.shape
        elif dxp is not None:
            dx = dxp
        if ctx.needs_input_grad[2] and not weight_gradients_disabled:
            if torch.is_grad_enabled():             # a graph of this backward pass is being recorded: keep everything differentiable
                xb = up._Upfirdn2dHip.apply(x, f, (1, 1, 1, 1, px0, px1, py0, py1, flip, ugain))
                dw = _ConvWgradHip.apply(dz, xb, cfg, tuple(w.shape), w.dtype, None)
            else:
                dw = _launch_wgrad_pieces(pieces, dz, cfg, tuple(w.shape), (bound, shape), w.dtype)
        if want_db and db is None:
            db = ba._BiasSum.apply(dz, 1)
        dres = dz if ctx.needs_input_grad[7] else None
        return dx, None, dw, db, None, None, None, dres, None

def blur_conv2d_s2_bias_act(x, f, weight, pad4, flip_filter=False, bias=None, act='linear', alpha=None, gain=None, clamp=None, wgain=1.0,
                            residual=None, passthrough=False):
    """``bias_act(conv2d(upfirdn2d(x, f, padding=pad4, flip_filter=flip_filter), weight, stride=2) [+ residual], bias, ...)`` with the blurred
    tensor as producer-written operand pieces (``pieces_available`` says whether the kernels exist for these shapes)."""
    from . import bias_act as ba
    spec = ba.activation_funcs[act]
    alpha = float(alpha if alpha is not None else spec.def_alpha)
    gain = float(gain if gain is not None else spec.def_gain)
    clampf = float(clamp if clamp is not None else -1)
    assert act in FUSABLE_ACTS
    cfg = _Cfg((False, 2, 0, 0, 0, 0, 1, float(wgain)))
    ucfg = (int(pad4[0]), int(pad4[1]), int(pad4[2]), int(pad4[3]), bool(flip_filter), 1.0)
    if passthrough and torch.is_grad_enabled() and x.requires_grad:
        y, again = _BlurConvS2Hip.apply(x, f, weight, bias, ucfg, cfg, (act, alpha, gain, clampf), residual, True)
        hit = getattr(x, '_pasta_amax', None)
        if hit is not None and not again.is_inference():
            again._pasta_amax = hit
        return y, again
    y = _BlurConvS2Hip.apply(x, f, weight, bias, ucfg, cfg, (act, alpha, gain, clampf), residual)
    return (y, x) if passthrough else y

_LINEAR_EPILOGUE = ('linear', 0.0, 1.0, -1.0)      # (act, alpha, gain, clamp) of an epilogue that only adds the residual

FUSABLE_ACTS = ('linear', 'relu', 'lrelu')

_cat1x1_cache = {}

def cat1x1_available(x, x2, weight):
    """Will ``conv2d_cat1x1_bias_act`` run as ONE launch over the two tensors (the pointwise kernel, pasta_conv2d_plan kernel 9)?"""
    if not (x.device.type == 'cuda' and x.dtype == torch.float32 and x2.dtype == torch.float32 and x.ndim == 4 and x2.shape[0] == x.shape[0]
            and x2.shape[2:] == x.shape[2:] and tuple(weight.shape[2:]) == (1, 1) and weight.shape[1] == x.shape[1] + x2.shape[1]
            and conv_math in ('default', 'f16x3') and x.numel() > 0):
        return False
    key = (int(x.shape[0]), int(x.shape[1]), int(x2.shape[1]), int(x.shape[2]), int(x.shape[3]), int(weight.shape[0]))
    hit = _cat1x1_cache.get(key)
    if hit is None:
        n, c1, c2, h, w_, o = key
        desc = _desc(_Cfg((False, 1, 0, 0, 0, 0, 1, 1.0)), (n, c1 + c2, h, w_), o, h, w_, 1, 1)
        kernel = ctypes.c_int()
        hit = _native.lib().pasta_conv2d_plan(ctypes.byref(desc), 4, None, None, None, None, ctypes.byref(kernel)) == 0 and kernel.value == 9
        _cat1x1_cache[key] = hit
    return hit

class _CatConv1x1BiasActHip(torch.autograd.Function):
    """``bias_act(conv2d(cat([x, x2], 1), w), b)`` for a 1x1 weight in one launch over the two tensors; the backward is assembled from the
    stand-alone differentiable pieces -- the two input gradients are two pointwise launches on the weight's channel slices (two
    contiguous tensors, where the concatenation's backward hands out channel slices of one), the weight gradient two launches whose
    results are concatenated (a few hundred KB)."""
    @staticmethod
    def forward(ctx, x, x2, w, b, wgain, act_cfg):
        act, alpha, gain, clamp = act_cfg
        from . import bias_act as ba
        cfg = _Cfg((False, 1, 0, 0, 0, 0, 1, float(wgain)))
        used = {}
        y = _launch_conv(x, w, cfg, epilogue=(b, ba.activation_funcs[act].cuda_idx, alpha, gain, clamp, None), x2=x2, used=used)
        keep_y = act != 'linear' or clamp >= 0
        ctx.save_for_backward(x, x2, w, b, y if keep_y else None)
        ctx.cfg, ctx.act_cfg, ctx.amax = cfg, act_cfg, (used.get('x_amax'), used.get('x2_amax'))
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import bias_act as ba
        x, x2, w, b, y = ctx.saved_tensors
        act, alpha, gain, clamp = ctx.act_cfg
        cfg = ctx.cfg
        dz = dy
        dx = dx2 = dw = db = None
        want_db = b is not None and ctx.needs_input_grad[3]
        if act != 'linear' or gain != 1 or clamp >= 0:
            if want_db:
                dz, db = ba.grad_with_bias_grad(dy.contiguous(), y, (1, act, alpha, gain, clamp))
            else:
                dz = ba._BiasActHipGrad.apply(dy.contiguous(), None, None, y, (1, act, alpha, gain, clamp))
        c1 = x.shape[1]
        gcfg = _grad_cfg(cfg, x.shape[2:], dz.shape[2:], 1, 1)
        if ctx.needs_input_grad[0]:
            dx = _ConvHip.apply(dz, w[:, :c1], gcfg)
        if ctx.needs_input_grad[1]:
            dx2 = _ConvHip.apply(dz, w[:, c1:], gcfg)
        if ctx.needs_input_grad[2] and not weight_gradients_disabled:
            dw = torch.cat([_ConvWgradHip.apply(dz, x, cfg, (w.shape[0], c1, 1, 1), w.dtype, ctx.amax[0]),
                            _ConvWgradHip.apply(dz, x2, cfg, (w.shape[0], w.shape[1] - c1, 1, 1), w.dtype, ctx.amax[1])], dim=1)
        if want_db and db is None:
            db = ba._BiasSum.apply(dz, 1)
        return dx, dx2, dw, db, None, None

def conv2d_cat1x1_bias_act(x, x2, weight, bias=None, act='linear', alpha=None, gain=None, clamp=None, wgain=1.0):
    """``bias_act(conv2d(torch.cat([x, x2], 1), weight), bias, ...)`` for a 1x1 ``weight`` without the concatenated tensor
    (``cat1x1_available`` says whether the one-launch form exists for these shapes; otherwise the concatenation is formed)."""
    from . import bias_act as ba
    spec = ba.activation_funcs[act]
    alpha = float(alpha if alpha is not None else spec.def_alpha)
    gain = float(gain if gain is not None else spec.def_gain)
    clampf = float(clamp if clamp is not None else -1)
    if act in FUSABLE_ACTS and cat1x1_available(x, x2, weight):
        return _CatConv1x1BiasActHip.apply(x, x2, weight, bias, float(wgain), (act, alpha, gain, clampf))
    return conv2d_bias_act(torch.cat([x, x2], dim=1), weight, bias, act=act, alpha=alpha, gain=gain, clamp=clamp, wgain=wgain)

def conv2d_bias_act(input, weight, bias=None, stride=1, padding=0, groups=1, act='linear', alpha=None, gain=None, clamp=None, wgain=1.0,
                    residual=None, passthrough=False):
    """``bias_act(conv2d(input, weight) [+ residual], bias, act, alpha, gain, clamp)`` in one launch (fp32 GPU tensors,
    act in FUSABLE_ACTS); other cases run the ops separately.  ``passthrough=True`` returns ``(y, input')`` where ``input'`` is the
    input again, for its other consumers: see ``_ConvBiasActHip``."""
    from . import bias_act as ba
    spec = ba.activation_funcs[act]
    alpha = float(alpha if alpha is not None else spec.def_alpha)
    gain = float(gain if gain is not None else spec.def_gain)
    clampf = float(clamp if clamp is not None else -1)
    if act in FUSABLE_ACTS and input.dtype in IO_CODES and input.device.type == 'cuda' and input.numel() > 0:
        sh, sw = _pair(stride)
        ph, pw = _pair(padding)
        assert sh == sw
        cfg = _Cfg((False, sh, ph, pw, 0, 0, int(groups), float(wgain)))
        if passthrough and torch.is_grad_enabled() and input.requires_grad:
            y, again = _ConvBiasActHip.apply(input, weight, bias, cfg, (act, alpha, gain, clampf), residual, True)
            hit = getattr(input, '_pasta_amax', None)      # the view autograd made of the input is the same data at the same version: its maxima go along
            if hit is not None and not again.is_inference():
                again._pasta_amax = hit
            hit = _shared_lookup(input)                     # ... and its packed pieces (share_pieces), for the next reader
            if hit is not None and not again.is_inference():
                again._pasta_pieces = (again._version, again.data_ptr()) + hit[2:]
            return y, again
        y = _ConvBiasActHip.apply(input, weight, bias, cfg, (act, alpha, gain, clampf), residual)
        return (y, input) if passthrough else y
    y = conv2d(input, weight, stride=stride, padding=padding, groups=groups, wgain=wgain)
    if residual is not None:
        y = y + residual
    y = ba.bias_act(y, bias, act=act, alpha=alpha, gain=gain, clamp=clamp)
    return (y, input) if passthrough else y

def modulated_conv2d_forward(x, weight, styles, dcoefs=None, stride=1, padding=0, transposed=False, per_sample=False, tail=None):
    """FORWARD-ONLY modulated convolution (networks.py:36-94) in one launch -- no autograd graph is recorded; the training
    path keeps its differentiable pieces.  ``x`` [N,I,H,W], ``weight`` [O,I,kh,kw] ([I,O,kh,kw] when ``transposed``),
    ``styles`` [N,I], ``dcoefs`` [N,O] or None.

    ``per_sample=False``: ``conv(x * styles) * dcoefs`` with the shared weight.  fp32 tensors: the styles ride in the
    kernel's activation staging (``iscale``), no modulated copy of ``x`` is written; 16-bit tensors are scaled by a pass of
    their own first.  ``per_sample=True``: the reference's grouped form (groups = N), each sample's weights
    ``w * styles[n] * dcoefs[n]`` formed by the weight-packing kernel.
    ``tail`` = dict(noise=unit plane(s) or None, strength=scalar tensor, bias, act, alpha, gain, clamp): the rest of
    SynthesisLayer / ToRGBLayer -- ``+ noise * strength``, bias, activation, gain, clamp -- in the convolution's epilogue
    (act in FUSABLE_ACTS).  Without ``tail`` the plain (for ``per_sample``: demodulated) convolution is returned and
    ``dcoefs`` of the shared form must be applied by the caller (an upsampling layer filters first)."""
    from . import bias_act as ba
    from . import fma
    assert not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, weight, styles, dcoefs))), \
        'modulated_conv2d_forward records no graph'
    n, o = int(x.shape[0]), int(weight.shape[1] if transposed else weight.shape[0])
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    assert sh == sw
    epilogue = noise = None
    if tail is not None:
        spec = ba.activation_funcs[tail['act']]
        assert tail['act'] in FUSABLE_ACTS
        bias = tail.get('bias')
        if bias is not None and per_sample:
            bias = bias.repeat(n)                       # the grouped launch has N * O output channels
        epilogue = (bias, spec.cuda_idx, float(tail['alpha'] if tail.get('alpha') is not None else spec.def_alpha),
                    float(tail['gain'] if tail.get('gain') is not None else spec.def_gain),
                    float(tail['clamp'] if tail.get('clamp') is not None else -1))
        if tail.get('noise') is not None:
            noise = (tail['noise'], tail['strength'])
    if per_sample:
        if noise is not None and noise[0].numel() != noise[0].shape[-1] * noise[0].shape[-2]:
            raise NotImplementedError('modulated_conv2d_forward: per-sample noise with per-sample weights')
        cfg = _Cfg((bool(transposed), sh, ph, pw, 0, 0, n, 1.0))
        y = _launch_conv(x.reshape(1, -1, *x.shape[2:]), weight, cfg, epilogue=epilogue, wmod=(styles, dcoefs), noise=noise)
        return y.reshape(n, o, *y.shape[2:])
    cfg = _Cfg((bool(transposed), sh, ph, pw, 0, 0, 1, 1.0))
    if x.dtype != torch.float32:
        x, styles = fma.scale_planes(x, styles), None
    if tail is None:
        return _launch_conv(x, weight, cfg, iscale=styles)
    return _launch_conv(x, weight, cfg, iscale=styles, oscale=dcoefs, epilogue=epilogue, noise=noise)

#----------------------------------------------------------------------------
# Shared-weight modulated convolution of the TRAINING path, y = conv(x * s[n, i], w)  (networks.py:72-76), without the tensor x * s:
#   forward          conv(x, w) with the styles multiplied onto the activations in the kernel's staging (``iscale``),
#   input gradient   conv^T(dy, w) with the styles as the output scale of the launch's epilogue (``oscale``),
#   weight gradient  the plain weight-gradient kernels on the UNMODULATED x with sample-aligned K slices, whose reduction forms
#                    dw = sum_n s[n, i] Dw_n and ds[n, i] = sum_{o, taps} w Dw_n  (pasta_conv2d_wgrad_modulated),
# instead of scale_planes (a read and a write of the activation) + a scan of x * s for its operand scale + scale_planes of the input
# gradient + plane_dot(dx, x) (two reads): seven passes over the activation per modulated layer.  

_MODCONV = _os.environ.get('PASTA_MODCONV_TRAIN', '1') != '0'         # A/B switch: 0 = scale_planes + the plain convolution, as before
_modconv_cache = {}

def modconv_available(x, weight, styles, stride=1, padding=0, transposed=False):
    """Does ``modulated_conv2d_shared`` run natively for these operands (else the caller scales the planes itself)?"""
    if not (_MODCONV and x.device.type == 'cuda' and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.ndim == 4 and x.numel() > 0
            and conv_math in ('default', 'f16x3') and not transposed and stride == 1):
        return False
    ph, pw = _pair(padding)
    key = (tuple(x.shape), tuple(weight.shape), ph, pw)
    hit = _modconv_cache.get(key)
    if hit is None:
        cfg = _Cfg((False, 1, ph, pw, 0, 0, 1, 1.0))
        kh, kw = int(weight.shape[2]), int(weight.shape[3])
        oh, ow = _out_hw(cfg, x.shape[2], x.shape[3], kh, kw)
        desc = _desc(cfg, x.shape, int(weight.shape[0]), oh, ow, kh, kw, kind='wgrad')
        hit = _native.lib().pasta_conv2d_wgrad_modulated_workspace(ctypes.byref(desc)) >= 0
        _modconv_cache[key] = hit
    return hit

def _launch_wgrad_modulated(x, dy, cfg, styles, w, x_amax=None):
    """(dw, dstyles) of ``conv(x * styles, w)`` given dy: pasta_conv2d_wgrad_modulated."""
    _native.require_gpu(x, 'conv2d_wgrad_modulated')
    kh, kw = int(w.shape[2]), int(w.shape[3])
    x, dy = _f32(x).contiguous(), _f32(dy).contiguous()
    styles, w32 = _f32(styles).contiguous(), _f32(w).contiguous()
    desc = _desc(cfg, x.shape, dy.shape[1], dy.shape[2], dy.shape[3], kh, kw, kind='wgrad')
    lib = _native.lib()
    nbytes = lib.pasta_conv2d_wgrad_modulated_workspace(ctypes.byref(desc))
    if nbytes < 0:
        raise RuntimeError('conv2d_wgrad_modulated: no sample-aligned weight-gradient kernel for this shape (modconv_available tells beforehand)')
    work = torch.empty([max(nbytes // 4, 4)], dtype=torch.float32, device=x.device)
    dw = torch.empty(list(w.shape), dtype=torch.float32, device=x.device)
    ds = torch.empty([x.shape[0], x.shape[1]], dtype=torch.float32, device=x.device)
    amax_x, amax_dy = (x_amax if x_amax is not None else tensor_amax(x)), tensor_amax(dy)
    desc.x_amax, desc.dy_amax = amax_x.data_ptr(), amax_dy.data_ptr()
    def launch():
        with torch.cuda.device(x.device):
            _native.check(lib.pasta_conv2d_wgrad_modulated(_native.ptr(x), _native.ptr(dy), _native.ptr(styles), _native.ptr(w32), _native.ptr(dw), _native.ptr(ds),
                                                           ctypes.byref(desc), _native.ptr(work), work.numel() * 4, _native.stream()))
    if launch_hook is None:
        launch()
    else:
        launch_hook('wgrad', desc, launch, 0)
    return dw, ds

class _ModConvHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, s, cfg):
        used = {}
        y = _launch_conv(x, w, cfg, iscale=s, used=used)
        ctx.save_for_backward(x, w, s)
        ctx.cfg, ctx.x_amax = cfg, used.get('x_amax')
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w, s = ctx.saved_tensors
        cfg = ctx.cfg
        dx = dw = ds = None
        dy = dy.contiguous()
        if ctx.needs_input_grad[0]:
            dx = _launch_conv(dy, w, _grad_cfg(cfg, x.shape[2:], dy.shape[2:], w.shape[2], w.shape[3]), oscale=s)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw, ds = _launch_wgrad_modulated(x, dy, cfg, s, w, x_amax=ctx.x_amax)
            dw = dw.to(w.dtype) if ctx.needs_input_grad[1] and not weight_gradients_disabled else None
            ds = ds.to(s.dtype) if ctx.needs_input_grad[2] else None
        return dx, dw, ds, None

def modulated_conv2d_shared(x, weight, styles, padding=0):
    """``conv2d(x * styles[:, :, None, None], weight, padding=padding)`` for the training path (where ``modconv_available``); first derivatives only,
    like the reference's own fused op would be used (the generator has no second-order phase: networks.py / loss_wo_flow_fullbody.py run no
    path-length regularisation).  ``styles``: [N, C_in]."""
    ph, pw = _pair(padding)
    cfg = _Cfg((False, 1, ph, pw, 0, 0, 1, 1.0))
    return _ModConvHip.apply(x, weight, styles.to(torch.float32).reshape(x.shape[0], x.shape[1]), cfg)

def demod_coefs(weight, styles):
    """``rsqrt(sum_{i,kh,kw} (weight[o,i] * styles[n,i])^2 + 1e-8)`` [N, O] (networks.py:65-68) by ``pasta_demod_coefs``: one
    workgroup per output channel, wavefront-shuffle reduction; differentiable to any order (the backward is written in
    differentiable torch operations on the small [N,I] / [O,I] matrices)."""
    return _DemodCoefs.apply(weight, styles)

class _DemodCoefs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, styles):
        _native.require_gpu(weight, 'demod_coefs')
        o, i = int(weight.shape[0]), int(weight.shape[1])
        n = int(styles.shape[0])
        assert styles.shape == (n, i)
        w32, s32 = _f32(weight).contiguous(), _f32(styles).contiguous()
        d = torch.empty([n, o], dtype=torch.float32, device=weight.device)
        with torch.cuda.device(weight.device):
            _native.check(_native.lib().pasta_demod_coefs(_native.ptr(w32), _native.ptr(s32), _native.ptr(d), n, o, i,
                                                          int(weight[0, 0].numel()), 1e-8, _native.stream()))
        ctx.save_for_backward(weight, styles, d)
        return d.to(styles.dtype)

    @staticmethod
    def backward(ctx, g):
        weight, styles, d = ctx.saved_tensors
        if torch.is_grad_enabled():         # a graph of this backward is being recorded: d has to be a function of the inputs again
            d = _DemodCoefs.apply(weight, styles).to(torch.float32)
        # d = q^-1/2, q[n,o] = sum_i s[n,i]^2 W2[o,i] + eps, W2 = sum_taps w^2
        dq = -0.5 * g.to(torch.float32) * d * d * d
        ds = dw = None
        if ctx.needs_input_grad[1]:
            ds = (2 * styles * (dq @ weight.square().sum(dim=[2, 3]).to(torch.float32))).to(styles.dtype)
        if ctx.needs_input_grad[0]:
            dw = (2 * weight * (dq.t() @ styles.square().to(torch.float32))[:, :, None, None]).to(weight.dtype)
        return dw, ds

class _ConvWgradHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, cfg, w_shape, w_dtype=None, x_amax=None):
        dw = _launch_wgrad(x, dy, cfg, w_shape, w_dtype, x_amax=x_amax)
        ctx.save_for_backward(dy, x)
        ctx.cfg = cfg
        return dw

    @staticmethod
    def backward(ctx, d_dw):
        dy, x = ctx.saved_tensors
        cfg = ctx.cfg
        d_dy = d_x = None
        if ctx.needs_input_grad[0]:     # y is linear in w: d_dy = conv(x, d_dw)
            d_dy = _ConvHip.apply(x, d_dw, cfg)
            assert d_dy.shape == dy.shape
        if ctx.needs_input_grad[1]:     # and dx is linear in w as well
            gcfg = _grad_cfg(cfg, x.shape[2:], dy.shape[2:], d_dw.shape[2], d_dw.shape[3])
            d_x = _ConvHip.apply(dy, d_dw, gcfg)
            assert d_x.shape == x.shape
        return d_dy, d_x, None, None, None, None

#----------------------------------------------------------------------------

def _check_common(input, weight, dilation):
    assert isinstance(input, torch.Tensor) and isinstance(weight, torch.Tensor)
    if _pair(dilation) != (1, 1):
        raise NotImplementedError('conv2d_gradfix: dilation != 1 is not on the PASTA-GAN path and is not implemented')

def _add_bias(y, bias):
    return y if bias is None else y + bias.to(y.dtype).reshape(1, -1, 1, 1)

def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, wgain=1.0):
    """Same contract as ``torch.nn.functional.conv2d`` (reference :35-38); equal strides in x and y.
    ``wgain`` (extension): convolve with ``weight * wgain`` without materialising the product."""
    _check_common(input, weight, dilation)
    sh, sw = _pair(stride)
    if sh != sw:
        raise NotImplementedError('conv2d_gradfix: anisotropic stride is not implemented')
    ph, pw = _pair(padding)
    cfg = _Cfg((False, sh, ph, pw, 0, 0, int(groups), float(wgain)))
    return _add_bias(_ConvHip.apply(input, weight, cfg), bias)

def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1, wgain=1.0):
    """Same contract as ``torch.nn.functional.conv_transpose2d`` (reference :40-43)."""
    _check_common(input, weight, dilation)
    sh, sw = _pair(stride)
    if sh != sw:
        raise NotImplementedError('conv2d_gradfix: anisotropic stride is not implemented')
    ph, pw = _pair(padding)
    oph, opw = _pair(output_padding)
    cfg = _Cfg((True, sh, ph, pw, oph, opw, int(groups), float(wgain)))
    return _add_bias(_ConvHip.apply(input, weight, cfg), bias)

#----------------------------------------------------------------------------
