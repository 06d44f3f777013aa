"""Benchmark: PASTA-GAN 256x192 (tensor 256x256) training images/sec on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Started through ``torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE in the environment)
each process is one rank; started plainly, this process only spawns the N ranks itself -- before any GPU call, like the
reference's launcher (train_wo_flow_fullbody.py:393-400, 567) -- waits for them and relays rank 0's JSON line.

A step = one iteration of the reference hot loop (training_loop_wo_flow_fullbody.py:484-529) on a
synthetic, HBM-resident batch of 16 images per GPU: Gmain + Dmain every iteration, Dreg every 16th,
Greg (a no-op besides the style encoder, pl_weight 0) every 4th, Adam steps and the EMA update.
Prints ONE JSON line (rank 0) with the whole-job throughput, the roofline of the dominant kernel
(measured with HIP events on the launch stream, inside the timed region) and a CPU baseline
(the oracle restatement on the host cores, config 1 of BASELINE.json).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, 'pasta-gan_amd'), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 matrix peak
PEAK_HBM_GBS = 8000.0
# kernel names as rocprofv3 prints them (without blanks); the last template argument = bf16 pieces per operand
# (3 = the default six-product arithmetic, 2 = bf16x3, 1 = bf16)
STORAGE_IO = 0        # pasta_conv_desc.io_dtype of the run (0 f32, 1 f16, 3 bf16): part of the kernel names, sets the product count
def _np():
    from torch_utils.ops import conv2d_gradfix
    return 1 if STORAGE_IO else {'bf16x3': 2, 'bf16': 1, 'bf16x6': 3}.get(conv2d_gradfix.conv_math, 4)     # 4 = NP_F16X3: 'default' / 'f16x3'
def bf16x6_names():         # template arguments as rocprofv3 prints them: <BM, BN, OCC, pieces, storage, input scale in the staging>
    return {0: f'conv_fwd_bf16x6_kernel<128,128,3,{_np()},{STORAGE_IO},false,false>', 1: f'conv_fwd_bf16x6_kernel<64,256,2,{_np()},{STORAGE_IO},false,false>'}
def bf16x6_packed_names():  # the few-input-channel (packed-K) mode: last template argument
    return {0: f'conv_fwd_bf16x6_kernel<128,128,3,{_np()},0,false,true>', 1: f'conv_fwd_bf16x6_kernel<64,256,2,{_np()},0,false,true>'}
def bf16x6_rows_names():    # <BM, BN, OCC, schedule, pieces, storage, input scale, parity pairs>
    return {0: f'conv_fwd_rows_bf16x6_kernel<128,128,2,1,{_np()},{STORAGE_IO},false,false>',
            1: f'conv_fwd_rows_bf16x6_kernel<64,256,2,1,{_np()},{STORAGE_IO},false,false>'}
def bf16x6_pair_names():    # stride-2 conv_transpose2d on the row-reuse kernel's parity-pair mode (+ its remainder launch)
    return {0: f'conv_fwd_rows_bf16x6_kernel<128,128,2,2,{_np()},0,false,true>', 1: f'conv_fwd_rows_bf16x6_kernel<64,256,2,2,{_np()},0,false,true>'}

TILE_NAMES = {0: 'conv_fwd_kernel<128,128,2,2,8,4>', 1: 'conv_fwd_kernel<64,256,2,2,8,4>',
              2: 'conv_fwd_kernel<32,256,1,2,8>', 3: 'conv_fwd_kernel<64,64,1,1,8>'}
WGRAD_NAMES = {0: 'conv_wgrad_kernel', 1: 'conv_wgrad_smallcin_kernel', 2: 'conv_wgrad3x3_bf16x6_kernel', 3: 'conv_wgrad3x3s2_bf16x6_kernel', 4: 'conv_wgrad1x1_bf16x6_kernel', 5: 'wgrad1x1_fewcin_kernel', 6: 'conv_wgrad3x3s2_pieces_kernel'}


class ConvMeter:
    """Brackets native convolution launches with HIP events on the launch stream and accumulates algorithmic FLOPs
    per kernel family.  ``only`` = None brackets every launch (the survey iteration); a set of family names brackets
    just those (the timed region: the dominant family, so that the event records do not slow the step down)."""

    def __init__(self, lib):
        self.lib = lib
        self.records = []          # (family, flops, kernels, start_event, end_event)
        self.enabled = False
        self.only = None
        self._family = {}          # descriptor fields -> (family, kernels): one plan call per distinct launch
        self.iteration = 0         # set by the caller before every step
        self.counts = {}           # iteration -> {family: kernel launches}: counted always (no events), for the PMC cross-check

    def __call__(self, kind, desc, launch, flags=0):
        import ctypes
        key = (kind, flags, desc.N, desc.C_in, desc.H, desc.W, desc.C_out, desc.OH, desc.OW, desc.kh, desc.kw, desc.stride, desc.pad_h, desc.pad_w,
               desc.groups, desc.transposed, desc.math, desc.io_dtype, desc.x_layout)
        hit = self._family.get(key)
        if hit is not None:
            c = self.counts.setdefault(self.iteration, {})
            c[hit[0]] = c.get(hit[0], 0) + hit[1]
            if not self.enabled or (self.only is not None and hit[0] not in self.only):
                return launch()
        g = desc.groups
        macs = desc.N * desc.C_out * (desc.C_in // g) * desc.kh * desc.kw
        macs *= (desc.H * desc.W) if desc.transposed else (desc.OH * desc.OW)
        if kind in ('conv', 'conv_isc'):
            isc = kind == 'conv_isc'           # forward-only modulated convolutions: the styles ride in the kernel's staging
            tile, ksplit, math, launches, kernel = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            self.lib.pasta_conv2d_plan(ctypes.byref(desc), int(flags) | int(isc), ctypes.byref(tile), ctypes.byref(ksplit), ctypes.byref(math), ctypes.byref(launches),
                                       ctypes.byref(kernel))
            family = {0: TILE_NAMES, 1: bf16x6_names(), 2: bf16x6_rows_names(), 3: bf16x6_pair_names(), 8: bf16x6_packed_names(),
                      4: {0: f'conv_fwd_rows2d_bf16x6_kernel<128,128,4,{_np()},{STORAGE_IO},false,256,false,false>'}, 5: {0: f'conv_fwd_rows2d_bf16x6_kernel<128,128,2,{_np()},{STORAGE_IO},false,256,false,false>'},
                      6: {1: f'conv_fwd_rows2d_bf16x6_kernel<64,256,8,{_np()},{STORAGE_IO},false,256,false,false>'},
                      7: {0: f'conv_fwd_rows2d_bf16x6_kernel<128,256,8,{_np()},0,false,512,false,false>'},      # <BM, BN, rows, pieces, storage, input scale, threads, x as pieces, weights by LDS-DMA>
                      9: {0: 'conv1x1_f16x3_kernel<128,128>', 1: 'conv1x1_f16x3_kernel<64,256>'},
                      10: {0: 'conv3x3s2_f16x3_kernel<128>', 1: 'conv3x3s2_f16x3_kernel<64>'},
                      11: {t: 'conv1x1_fewcin_kernel' for t in range(4)}, 12: {t: 'conv1x1_fewcout_kernel' for t in range(4)},
                      13: {t: 'conv_t2_f16x3_kernel<false,%d>' % (32 if (desc.H % 8 == 0 and desc.W % 32 == 0) else 16) for t in range(4)}}[kernel.value][tile.value]     # <input scale, tile columns>     # <BM, BN, rows per tile, pieces, storage, input scale, threads>
            if kernel.value == 10 and desc.x_layout:     # round 5: x as the producer wrote it (the blur's operand pieces): another instance
                family = family.replace('>', ',true>')
            if kernel.value == 7 and desc.x_layout:      # ... and the eight-wave tile reading pieces (the SPADE feature map)
                family = family.replace(',false,false>', ',true,false>')
            if isc and kernel.value == 13:
                family = family.replace('<false,', '<true,')
            elif isc and kernel.value != 0:       # the instance with the input scale: same family, another template argument (rocprofv3 names)
                family = {1: family.replace(',3,3,0,false,false>', ',2,3,0,true,false>').replace(',3,4,0,false,false>', ',2,4,0,true,false>'), 2: family.replace(',false,false>', ',true,false>')}.get(
                    kernel.value, family.replace(',false,256,false,false>', ',true,256,false,false>').replace(',false,512,false,false>', ',true,512,false,false>'))      # (round 4: the eight-wave tile has its instance too)
            kernels = launches.value
        else:
            which = ctypes.c_int()
            self.lib.pasta_conv2d_wgrad_plan(ctypes.byref(desc), ctypes.byref(which))
            family = WGRAD_NAMES[which.value]
            kernels = 1
        if hit is None:
            self._family[key] = (family, kernels)
            c = self.counts.setdefault(self.iteration, {})
            c[family] = c.get(family, 0) + kernels
        if not self.enabled or (self.only is not None and family not in self.only):
            return launch()
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        launch()
        e.record()
        shape = (kind, desc.N, desc.C_in, desc.H, desc.C_out, desc.OH, desc.kh, desc.stride, desc.transposed, desc.groups)
        self.records.append((family, 2.0 * macs, kernels, s, e, shape))

    def summary(self):
        fam = {}
        for family, flops, kernels, s, e, _shape in self.records:
            f = fam.setdefault(family, dict(flops=0.0, ms=0.0, launches=0, kernels=0))
            f['flops'] += flops
            f['ms'] += s.elapsed_time(e)
            f['launches'] += 1
            f['kernels'] += kernels
        return fam


    def by_shape(self):
        """(kind, N, Cin, H, Cout, OH, k, stride, transposed, groups) -> [calls, total ms, TFLOP/s]"""
        tab = {}
        for family, flops, kernels, s, e, shape in self.records:
            t = tab.setdefault(shape, [0, 0.0, 0.0])
            t[0] += 1; t[1] += s.elapsed_time(e); t[2] += flops
        return sorted(((k, v[0], v[1], v[2] / (v[1] * 1e-3) / 1e12) for k, v in tab.items()), key=lambda r: -r[2])


def pmc_traffic(kernel_family, expected_launches):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes (profiles/r*_pmc.json: separate
    --pmc runs of ``bench.py --steps 1 --warmup 1 --no-cpu-baseline``, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950).  The figure is only reported when the profile is of THIS code: the kernel must have been launched there
    exactly as often as this run launches it in the same two iterations (iteration 0 with every phase + one plain
    iteration).  Returns (bytes per launch or None, file name or the reason for None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc.json')))
    if not files:
        return None, 'no profiles/r*_pmc.json'
    want = kernel_family.replace(' ', '')
    name_ = os.path.basename(files[-1])
    for name, v in json.load(open(files[-1])).items():
        if name.startswith('_'):
            continue
        if want in name.replace(' ', ''):
            if expected_launches is None:
                return None, f'{name_}: launch count of this run unknown (needs --warmup >= 2 with the meter on)'
            if int(v['launches']) != int(expected_launches):
                return None, (f'{name_} is stale: it holds {int(v["launches"])} launches of this kernel over two iterations, this code makes '
                              f'{int(expected_launches)}')
            return float(v['hbm_bytes_corrected']), name_
    return None, f'{name_}: kernel not in the profile'




def host_threads():
    """CPU threads this process may really use: the cgroup quota when there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(budget_s=20.0):
    """Config 1 of BASELINE.json on the host: batch 2, G fwd + D(img), D(finetune), D(real) fwd, GAN softplus
    losses, one backward into G and D -- the oracle restatement (same torch-op composition as the reference's
    CPU fallback), full 'fashion' widths, fp32."""
    from oracle import ref_networks as RN, ref_ops
    ref_ops.FIR_AS_DEPTHWISE_CONV = True      # the reference's own CPU formulation of the FIR step
    from training.training_loop_wo_flow_fullbody import fashion_config, SyntheticFullBodyBatch
    import dnnlib
    cfg = fashion_config()
    torch.manual_seed(0)
    G = dnnlib.util.construct_class_by_name(**cfg.G_kwargs)
    D = dnnlib.util.construct_class_by_name(**cfg.D_kwargs)
    sdG = {k: v.detach().requires_grad_(v.dtype.is_floating_point and k in dict(G.named_parameters())) for k, v in list(G.named_parameters()) + list(G.named_buffers())}
    sdD = {k: v.detach().requires_grad_(k in dict(D.named_parameters())) for k, v in list(D.named_parameters()) + list(D.named_buffers())}
    data = SyntheticFullBodyBatch(2, torch.device('cpu'), seed=0).tensors
    softplus = torch.nn.functional.softplus

    def step():
        img, fin, par = RN.generator_full(sdG, torch.zeros([2, 0]), data['style_input'], data['retain'], data['pose'],
                                          data['denorm_upper_input'], data['denorm_lower_input'], data['denorm_upper_mask'],
                                          data['denorm_lower_mask'], conv_clamp=256, mapping_layers=1, noise_mode='const')
        c = torch.tanh(data['style_input'].mean(dim=[2, 3]).repeat(1, 13)[:, :512])
        loss = softplus(-RN.discriminator(sdD, img, c)).mean() + softplus(-RN.discriminator(sdD, fin, c)).mean() + \
            softplus(-RN.discriminator(sdD, data['real_img'], c)).mean()
        params = [v for v in list(sdG.values()) + list(sdD.values()) if v.requires_grad]
        torch.autograd.grad(loss, params, allow_unused=True)

    cores = host_threads()
    torch.set_num_threads(cores)
    step()                                   # warm-up
    t0 = time.time()
    n = 0
    while n < 3 and (n == 0 or time.time() - t0 < budget_s * 0.6):
        step()
        n += 1
    dt = time.time() - t0
    return dict(value=round(2 * n / dt, 4), unit='images/sec', cores=cores, kind='port',
                sample=f'BASELINE config 1: batch 2, G fwd + 3x D fwd + one backward, fp32, full fashion widths, '
                       f'{n} timed step(s) of {dt / n:.2f} s after 1 warm-up (oracle/ref_networks.py on the host)')


class BytesMeter:
    """HIP-event brackets around upfirdn2d launches (on the launch stream): algorithmic bytes and time per shape."""
    def __init__(self):
        self.records, self.enabled = [], False

    def __call__(self, nbytes, key, launch):
        if not self.enabled:
            return launch()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); launch(); e.record()
        self.records.append((key, nbytes, s, e))

    def table(self):
        tab = {}
        for key, nbytes, s, e in self.records:
            t = tab.setdefault(key, [0, 0.0, 0.0])
            t[0] += 1; t[1] += s.elapsed_time(e); t[2] += nbytes
        return tab


def run_infer(args, device):
    """BASELINE config 4: generator inference, batch 8, the call sequence of test.py:120-128 / test_512.py:127-134."""
    import dnnlib  # noqa: F401
    from training import networks
    from training.training_loop_wo_flow_fullbody import SyntheticFullBodyBatch
    from torch_utils.ops import upfirdn2d, conv2d_gradfix
    res = args.res or 512
    batch = args.batch_gpu if args.batch_gpu != 16 else 8
    common = dict(z_dim=0, c_dim=512, w_dim=512, img_resolution=res, img_channels=3, mapping_kwargs=dict(num_layers=1),
                  synthesis_kwargs=dict(channel_base=16384, channel_max=512, conv_clamp=256))
    torch.manual_seed(0)
    if res == 256:
        G, patch_ch, cls_note = networks.GeneratorV18(**common), 60, "GeneratorV18 (test.py's released class; parity pinned by the reference fixture)"
    else:
        G, patch_ch, cls_note = networks.GeneratorFull(**common), 42, ('GeneratorFull generalised to 512 (test_512.py drives a class the reference does not '
                                                                       'ship; this is the package\'s own generalisation, parity UNPINNED)')
    G = G.eval().requires_grad_(False).to(device)
    data = SyntheticFullBodyBatch(batch, device, seed=0, res=res).tensors
    patches = data['style_input'].repeat(1, 2, 1, 1)[:, :patch_ch].contiguous()
    z = torch.zeros([batch, 0], device=device)

    def step():
        with torch.no_grad():
            code, pyramid = G.style_encoding(patches, data['retain'])
            pose_feat = G.const_encoding(data['pose'])
            ws = G.mapping(z, code)
            feats = {str(f.shape[2]): f for f in pyramid}
            return G.synthesis(ws, pose_feat, feats, data['denorm_upper_input'], data['denorm_lower_input'], data['denorm_upper_mask'],
                               data['denorm_lower_mask'], noise_mode='const')

    meter = BytesMeter()
    upfirdn2d.launch_hook = meter
    conv = ConvMeter(_native_lib())
    conv2d_gradfix.launch_hook = conv
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    meter.enabled = conv.enabled = not args.no_meter
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    meter.enabled = conv.enabled = False
    upfirdn2d.launch_hook = conv2d_gradfix.launch_hook = None
    assert all(torch.isfinite(t).all() for t in out[:2])
    graph_line = None
    if args.graph:
        # the same forward captured once as a hipGraph and replayed -- what the host's launch work is worth here.
        # Everything the forward allocates comes from torch's capture pool; the rows of producer-side maxima are zeroed INSIDE the graph
        # (torch_utils/ops/_native.py, amax_slot), so a replay starts from the same state as an eager call.
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()                                  # once on the capture stream (lazy initialisations, caches)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                gout = step()
            g.replay(); torch.cuda.synchronize()
            same = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(gout[:2], out[:2]))
            tg = time.perf_counter()
            for _ in range(args.steps):
                g.replay()
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            graph_line = {'value': round(args.steps * batch / tg, 2), 'unit': 'images/sec', 'ms_per_step': round(1000 * tg / args.steps, 2),
                          'max_rel_difference_from_eager': same, 'note': 'the timed eager forward captured once (torch.cuda.CUDAGraph = hipGraph) and replayed'}
        except Exception as e:  # noqa: BLE001  (a measurement, never the headline: report why it could not be taken)
            graph_line = {'error': f'{type(e).__name__}: {e}'[:400]}
    line = {'metric': f'generator inference images/sec at {res}x{res * 5 // 8 if res == 512 else 192} (tensor {res}x{res}), batch {batch}', 'value': round(args.steps * batch / dt, 2),
            'unit': 'images/sec', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1000 * dt / args.steps, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'conv_math': conv2d_gradfix.conv_math, 'data': 'synthetic',
            'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            'config': {'workload': f'BASELINE config 4: generator inference, eval mode, per-sample modulated weights (grouped convolution, groups = batch), '
                                   f'noise_mode const, cfg=fashion widths, random-init weights; {cls_note}', 'global_batch': batch, 'parallelism': 'dp1 (single process)'}}
    if graph_line is not None:
        line['hipgraph_replay'] = graph_line
    tab = meter.table()
    if tab:
        nbytes, ms = sum(v[2] for v in tab.values()), sum(v[1] for v in tab.values())
        big = max(tab.items(), key=lambda kv: kv[1][1])
        line['roofline'] = {'bound': 'hbm', 'kernel': 'upfirdn2d (all launches of the generator)', 'achieved': round(nbytes / (ms * 1e-3) / 1e9, 1), 'peak': PEAK_HBM_GBS,
                            'unit': 'GB/s', 'frac': round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), 'traffic': None,
                            'traffic_note': 'no PMC pass committed for the inference workload',
                            'algorithmic_bytes_per_step': round(nbytes / args.steps), 'launches_per_step': round(sum(v[0] for v in tab.values()) / args.steps, 1),
                            'share_of_step': round(ms / (1000 * dt), 3),
                            'largest_shape': {'shape': str(big[0]), 'gbps': round(big[1][2] / (big[1][1] * 1e-3) / 1e9, 1), 'ms_per_step': round(big[1][1] / args.steps, 3)}}
    fam = conv.summary()
    if fam:
        line['conv_families'] = {k: {'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2), 'ms_per_step': round(v['ms'] / args.steps, 2),
                                     'launches_per_step': round(v['launches'] / args.steps, 1)} for k, v in sorted(fam.items())}
        line['conv_total'] = {'tflop_per_step': round(sum(v['flops'] for v in fam.values()) / args.steps / 1e12, 3),
                              'ms_per_step': round(sum(v['ms'] for v in fam.values()) / args.steps, 2)}
    if args.by_shape:
        for key, (calls, ms, nb) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
            print(f'upfirdn2d {str(key):60s} calls/step={calls / args.steps:5.1f} ms/step={ms / args.steps:7.3f} GB/s={nb / (ms * 1e-3) / 1e9:7.1f}', file=sys.stderr)
        for shape, calls, ms, tf in conv.by_shape()[:args.by_shape_top]:
            print(f'{str(shape):70s} calls/step={calls / args.steps:6.1f} ms/step={ms / args.steps:8.2f} TF/s={tf:7.1f}', file=sys.stderr)
    print(json.dumps(line), flush=True)


def _native_lib():
    from torch_utils.ops import _native
    return _native.lib()


def launch_ranks(n, argv):
    """Start ``n`` rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), wait for
    them and relay rank 0's stdout.  This process makes no GPU call (``import torch`` and the hipcc build do not touch the
    device) and never re-executes itself; a rank that fails takes the others down and the exit status is non-zero."""
    import socket
    import subprocess
    import tempfile
    from torch_utils import custom_ops
    custom_ops.build()                      # once, before the ranks race for it
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode='w+')
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', str(max(1, host_threads() // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0 if r == 0 else sys.stderr))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        time.sleep(0.2)
        failed = next((p for p in procs if p.poll() not in (None, 0)), None)
    if failed is None:
        failed = next((p for p in procs if p.returncode != 0), None)
    if failed is not None:
        for p in procs:                     # exactly the processes started above, by PID
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        print(f'bench.py: rank {procs.index(failed)} exited with status {failed.returncode}', file=sys.stderr)
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    return 0 if failed is None else (failed.returncode or 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=16, help='timed iterations; 16 = one full lazy-regularisation period (Dreg runs every 16th)')
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch-gpu', type=int, default=16)
    ap.add_argument('--vgg-weight', type=float, default=0.0, help='> 0: add the VGG-19 perceptual term with random-init weights '
                    '(train.sh uses 40; the pretrained checkpoint cannot be obtained here). Not the headline configuration.')
    ap.add_argument('--aug', default='noaug', choices=['noaug', 'ada', 'fixed'], help="discriminator augmentation (SURVEY 8d: 'noaug' is the "
                    "headline; 'ada' = the shipped train.sh default, pipeline 'bgc', target 0.6)")
    ap.add_argument('--aug-p', type=float, default=0.5, help='initial (ada) or constant (fixed) augmentation probability; the reference starts '
                    'ADA at 0 and takes ~100 kimg to reach its working point, a benchmark has to start near it')
    ap.add_argument('--conv-math', default=None, choices=['default', 'f32', 'bf16x6', 'bf16x3', 'bf16', 'f16x3'],
                    help="matrix-core arithmetic of the convolutions (default: PASTA_CONV_MATH or 'default' = f16x3: fp32-equivalent products from three fp16 MFMAs; "
                         "'bf16x6' = the six-product fp32-equivalent arithmetic, the default until round 3; 'f32' = fp32 MFMA). "
                         "'bf16x3' = what TrainingStep selects for allow_tf32=True; 'bf16' = bf16 operands. Reduced modes are reported as such, never as the headline")
    ap.add_argument('--ddp-mode', default='flat', choices=['flat', 'torch'], help="gradient exchange at N > 1: 'flat' = one bucketed reducer per "
                    "optimised module (training/grad_reducer.py); 'torch' = the reference's five DistributedDataParallel wrappers")
    ap.add_argument('--mode', default='train', choices=['train', 'infer'], help="'train' (default, the headline): the full training step of BASELINE "
                    "config 2.  'infer': generator inference as test.py / test_512.py run it (BASELINE config 4): eval mode, per-sample weights "
                    "(fused_modconv, grouped convolution), const noise; reports generated images/sec and the HBM roofline of upfirdn2d")
    ap.add_argument('--res', type=int, default=None, choices=[256, 512], help="--mode infer: 256 = GeneratorV18 (test.py's class, parity pinned); "
                    "512 (default) = the resolution-generalised GeneratorFull standing in for test_512.py's unreleased class (parity unpinned)")
    ap.add_argument('--storage', default='f32', choices=['f32', 'bf16', 'f16'], help="activation storage (BASELINE config 5 = bf16): 16-bit tensors in "
                    "HBM for the generator's synthesis network and encoders and every discriminator block, one matrix-core product per multiply-add, "
                    "fp32 accumulation, demodulation, statistics and images.  Reduced precision: reported as such, never the headline")
    ap.add_argument('--train-res', type=int, default=256, choices=[256, 512], help='--mode train: 512 = the resolution-generalised model (config 5: 512x320, batch 8/GPU)')
    ap.add_argument('--d-fp16-res', type=int, default=0, help="discriminator blocks of the N highest resolutions store and multiply in fp16 "
                    "(networks.py:1107-1120).  The reference's train script sets 4 (train_wo_flow_fullbody.py:195-196); the headline keeps 0 = "
                    "everything fp32-equivalent, which is what the parity oracle (the reference's force_fp32 CPU path) computes")
    ap.add_argument('--no-variants', action='store_true', help='skip the two reduced-precision side measurements (fp16 discriminator blocks as in '
                    'the reference train script; bf16 activation storage = BASELINE config 5) that a default single-GPU run appends as `also_measured`')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='--mode infer: also capture the forward as a hipGraph and time its replay (reported as hipgraph_replay)')
    ap.add_argument('--no-meter', action='store_true', help='do not bracket convolution launches with events')
    ap.add_argument('--replica-check', action='store_true', help='N > 1: after the timed steps compare the replicas (bitwise parameter checksums over all ranks) '
                    'and report fp64 sums of the parameters and of the last averaged gradients, so that two transports can be compared (tests/test_ddp_gpu.py)')
    ap.add_argument('--by-shape', action='store_true', help='also print a per-shape convolution table to stderr')
    ap.add_argument('--by-shape-top', type=int, default=40, help='rows of that table')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    # One rank per GPU. PASTA_DIST_BACKEND=gloo lets several ranks share one card for a rehearsal on a 1-GPU box.
    backend = os.environ.get('PASTA_DIST_BACKEND', 'nccl')
    n_dev = torch.cuda.device_count()
    assert n_dev > 0, 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    if backend == 'nccl' and world > n_dev:
        raise SystemExit(f'--gpus {world} over RCCL needs {world} GPUs, this node shows {n_dev} '
                         f'(PASTA_DIST_BACKEND=gloo rehearses several ranks on one card)')
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    transport = 'single process'
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            torch.distributed.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=device)
        else:
            torch.distributed.init_process_group(backend=backend, rank=rank, world_size=world)
        # the in-tree library is built (if stale) by rank 0 only, then loaded by everyone
        if rank == 0:
            from torch_utils import custom_ops
            custom_ops.build()
        torch.distributed.barrier()
        used = torch.distributed.get_backend()
        transport = ('RCCL all-reduce over xGMI (torch.distributed backend nccl)' if used == 'nccl' else
                     f'{used} all-reduce, REHEARSAL transport: ranks share {n_dev} GPU(s)') + \
                    f', process-group world_size {torch.distributed.get_world_size()}, gradient exchange: {args.ddp_mode}'

    from torch_utils.ops import conv2d_gradfix, _native
    from training.training_loop_wo_flow_fullbody import TrainingStep, SyntheticFullBodyBatch, fashion_config

    lib = _native.lib()       # raises if libpasta_hip.so is missing
    if args.conv_math is not None:
        conv2d_gradfix.conv_math = args.conv_math
    if args.mode == 'infer':
        assert world == 1, '--mode infer is a single-GPU measurement (replicas only: nothing is exchanged)'
        return run_infer(args, device)
    act = {'f32': None, 'bf16': 'bfloat16', 'f16': 'float16'}[args.storage]
    global STORAGE_IO
    STORAGE_IO = {'f32': 0, 'bf16': 3, 'f16': 1}[args.storage]
    cfg = fashion_config(mbstd_group_size=min(args.batch_gpu, 4), d_fp16_res=args.d_fp16_res, img_resolution=args.train_res, act_dtype=act)      # train_wo_flow_fullbody.py:184: mbstd = min(batch_gpu, 4)
    from training.training_loop_wo_flow_fullbody import augment_options
    cfg.update(augment_options(aug=args.aug, augpipe='bgc', p=args.aug_p))
    if args.vgg_weight > 0:
        cfg.loss_kwargs.vgg_weight = args.vgg_weight
        cfg.loss_kwargs.vgg_random_init = True
    step = TrainingStep(device, cfg=cfg, num_gpus=world, rank=rank, batch_size=args.batch_gpu * world, batch_gpu=args.batch_gpu, ddp_mode=args.ddp_mode)
    data = SyntheticFullBodyBatch(args.batch_gpu, device, seed=rank, res=args.train_res)
    meter = ConvMeter(lib)
    if not args.no_meter:
        conv2d_gradfix.launch_hook = meter

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # The last warm-up iteration is the survey: every convolution launch is bracketed, which tells which kernel family
    # dominates; in the timed region only that family is bracketed (full bracketing costs about 2 % of the step).
    survey = None
    for it in range(args.warmup):
        meter.enabled = (not args.no_meter) and it == args.warmup - 1
        meter.iteration = it
        step.run(data)
    torch.cuda.synchronize()
    if meter.enabled:
        survey = meter.summary()
        if survey and not args.by_shape:
            meter.only = {max(survey.items(), key=lambda kv: kv[1]['ms'])[0]}
        meter.records = []
    barrier()
    meter.enabled = not args.no_meter
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.steps):
        meter.iteration = args.warmup + it
        step.run(data)
    host_dt = time.perf_counter() - t0        # when the host had issued everything (no sync inside the loop)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    meter.enabled = False
    conv2d_gradfix.launch_hook = None

    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())

    replica = None
    if args.replica_check:
        # replicas start from rank 0's weights and apply the same averaged gradients: after any number of steps their parameters are
        # bit-identical (training_loop_wo_flow_fullbody.py:316-324 checks the same with check_ddp_consistency at every snapshot)
        params = [q for m in (step.G, step.D) for q in m.parameters()]
        bits = torch.stack([q.detach().view(torch.int32).sum(dtype=torch.int64) for q in params]).sum().reshape(1)
        every = [torch.zeros_like(bits) for _ in range(world)]
        if world > 1:
            torch.distributed.all_gather(every, bits)
        else:
            every = [bits]
        grads = [q.grad for q in params if q.grad is not None]
        replica = {'bit_identical': all(int(e.item()) == int(bits.item()) for e in every), 'world_size': world,
                   'param_abs_sum': float(sum(q.detach().double().abs().sum() for q in params).item()),
                   'grad_abs_sum': float(sum(g_.double().abs().sum() for g_ in grads).item()) if grads else None, 'grads': len(grads)}

    if rank == 0:
        images = args.steps * args.batch_gpu * world
        out = {
            'metric': 'training images/sec at 256x192 (tensor 256x256), batch 16 per GPU' if args.train_res == 256 else
                      f'training images/sec at 512x320 (tensor 512x512), batch {args.batch_gpu} per GPU',
            'value': round(images / dt, 3), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1000 * dt / args.steps, 2), 'host_issue_ms_per_step': round(1000 * host_dt / args.steps, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'bf16x3': 'f32 storage, split-bf16 x3 products (reduced: allow_tf32 counterpart)', 'bf16': 'f32 storage, bf16 operands (reduced: mixed precision)'}.get(conv2d_gradfix.conv_math, 'f32') +
                     (f'; discriminator b256..b{256 >> (args.d_fp16_res - 1)} in fp16 storage and products (reduced: the reference train script\'s mixed precision)' if args.d_fp16_res > 0 else '') +
                     (f'; {args.storage} ACTIVATION STORAGE in G and D, one {args.storage} matrix-core product per multiply-add, fp32 accumulation / demodulation / statistics (reduced: BASELINE config 5, tolerance in tests/test_storage16_gpu.py)' if act else ''), 'conv_math': conv2d_gradfix.conv_math, 'data': 'synthetic', 'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            'config': {'workload': ('BASELINE config 2' if args.train_res == 256 and act is None else 'BASELINE config 5 (one GPU of it)' if act else 'config 2 at 512') +
                                   ': full training_loop step (Gmain+Dmain every iter, Dreg/16, Greg/4, Adam, EMA), '
                                   f'GeneratorFull+Discriminator cfg=fashion {args.train_res}x{args.train_res}, batch {args.batch_gpu}/GPU, ' + ('G and D in fp32, ' if args.d_fp16_res == 0 else f'G in fp32, D num_fp16_res={args.d_fp16_res}, ') +
                                   ('vgg_weight=0 (weights unavailable)' if args.vgg_weight <= 0 else f'vgg_weight={args.vgg_weight:g} with random-init VGG-19') + (', no ADA' if args.aug == 'noaug' else f', ADA pipeline bgc ({args.aug}, p0={args.aug_p:g})') + ', random-init weights; timed iterations ' + f'{args.warmup}..{args.warmup + args.steps - 1}',
                       'global_batch': args.batch_gpu * world, 'parallelism': f'dp{world} ({transport})'},
        }
        if replica is not None:
            out['replica_check'] = replica
        fam = meter.summary()
        if fam:
            dom = max(fam.items(), key=lambda kv: kv[1]['ms'])
            name, f = dom
            achieved = f['flops'] / (f['ms'] * 1e-3) / 1e12
            # launches of this kernel in iteration 0 (every phase) + one plain iteration = what the PMC command executes
            plain = next((i for i in sorted(meter.counts) if i > 0 and i % 4 != 0), None)
            expected = (meter.counts.get(0, {}).get(name, 0) + meter.counts[plain].get(name, 0)) if (0 in meter.counts and plain is not None) else None
            traffic, src = pmc_traffic(name, expected)
            if 'bf16x6' in name:        # every split-bf16 kernel family (forward-type base / row / pair / 2-D, the three weight gradients)
                # six (three, one) bf16 MFMA products per multiply-add: the matrix pipes execute that multiple of the algorithmic FLOPs
                nprod = 1 if STORAGE_IO else {'bf16x3': 3, 'bf16': 1, 'bf16x6': 6}.get(conv2d_gradfix.conv_math, 3)
                peak = PEAK_BF16_MFMA_TFLOPS / nprod
                f16x3 = not STORAGE_IO and conv2d_gradfix.conv_math in ('default', 'f16x3')
                note = ('%s products from %d x v_mfma_f32_32x32x16_%s, fp32 accumulate; peak = 2500 TFLOP/s '
                        'dense 16-bit MFMA / %d; executed 16-bit MFMA rate = %d x achieved = %.0f TFLOP/s = %.1f%% of 2.5 PFLOP/s' %
                        ('fp32-equivalent (PASTA_MATH_F16X3: fp16 hi / lo pieces of power-of-two-scaled operands)' if f16x3 else
                         {6: 'split-bf16: fp32-equivalent', 3: 'split-bf16: 2^-16-accurate (opt-in allow_tf32 counterpart)', 1: 'bf16-operand (opt-in mixed precision)'}[nprod],
                         nprod, 'f16' if f16x3 else 'bf16', nprod, nprod, nprod * achieved, 100 * nprod * achieved / PEAK_BF16_MFMA_TFLOPS))
            else:
                peak = PEAK_F32_MFMA_TFLOPS
                note = 'dense fp32-input MFMA (v_mfma_f32_32x32x2_f32), exact-f32 products'
            out['roofline'] = {'bound': 'mfma', 'kernel': name, 'achieved': round(achieved, 2), 'peak': round(peak, 1),
                               'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': traffic,
                               'traffic_source': src, 'algorithmic_flop_per_launch': round(f['flops'] / f['launches']),
                               'launches': f['launches'], 'kernels': f['kernels'],
                               'avg_kernel_us': round(1000 * f['ms'] / f['kernels'], 1),
                               'share_of_step': round(f['ms'] / (1000 * dt), 3), 'peak_note': note}
            # every family: from the survey iteration when there was one (a Gmain + Dmain iteration without the lazy
            # regularisation phases), else from the fully bracketed timed region
            allfam, nit, src = (survey, 1, 'survey: last warm-up iteration, every launch bracketed') if survey and meter.only is not None \
                else (fam, args.steps, 'timed region, every launch bracketed')
            tot_flops = sum(v['flops'] for v in allfam.values())
            tot_ms = sum(v['ms'] for v in allfam.values())
            out['conv_families'] = {k: {'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2), 'ms_per_step': round(v['ms'] / nit, 2),
                                        'launches_per_step': round(v['launches'] / nit, 1)} for k, v in sorted(allfam.items())}
            out['conv_total'] = {'tflop_per_step': round(tot_flops / nit / 1e12, 3), 'ms_per_step': round(tot_ms / nit, 2), 'source': src}
        if args.by_shape:
            for shape, calls, ms, tf in meter.by_shape()[:args.by_shape_top]:
                print(f'{str(shape):70s} calls/step={calls / args.steps:6.1f} ms/step={ms / args.steps:8.2f} TF/s={tf:7.1f}', file=sys.stderr)
        headline_cfg = args.mode == 'train' and act is None and args.d_fp16_res == 0 and args.train_res == 256 and args.conv_math in (None, 'default')
        if world == 1 and headline_cfg and not args.no_variants and args.aug == 'noaug' and args.vgg_weight <= 0:
            # Side measurements, AFTER the headline's timed region and never part of `value`: the same step at the precision the
            # reference's own train script uses for D, and in 16-bit activation storage (BASELINE config 5 at this resolution).
            del step
            torch.cuda.empty_cache()
            out['also_measured'] = {}
            math_before = conv2d_gradfix.conv_math          # the side lines set their own arithmetic; the caller's comes back afterwards
            for tag, kw, vbatch, vres, note in [('conv_math_bf16x6', dict(), args.batch_gpu, 256, 'the same step with the six-product split-bf16 arithmetic (PASTA_MATH_BF16X6: the default until round 3; '
                                                                        'fp32-equivalent as well, no operand scales, twice the matrix work)'),
                                                ('d_fp16_res_3', dict(d_fp16_res=3), args.batch_gpu, 256, 'D blocks b256..b64 in fp16 storage + products: num_fp16_res = 3, conv_clamp = 256 as train_wo_flow_fullbody.py:195-196 sets them '
                                                                        '(GeneratorFull forces its blocks to fp32, networks.py:2307,2331): the reference script\'s default precision; '
                                                                        'parity: reference fixture with num_fp16_res = 4, tests/test_fullwidth.py (2e-2)'),
                                  ('storage_bf16', dict(act_dtype='bfloat16'), args.batch_gpu, 256, 'bf16 activation storage in G and D (BASELINE config 5 arithmetic at 256x256, batch 16); '
                                                                               'parity: oracle in the same storage type, tests/test_storage16_gpu.py'),
                                  ('storage_bf16_512', dict(act_dtype='bfloat16', img_resolution=512), 8, 512, 'BASELINE config 5, one GPU of it: 512x320 (tensor 512x512) training step, batch 8, bf16 '
                                                                               'activation storage with fp32 demodulation / accumulation; the 512 model is the resolution-generalised GeneratorFull '
                                                                               '(parity UNPINNED: the reference ships no 512 class); parity vs the oracle in the same storage type: tests/test_config5_gpu.py')]:
                vcfg = fashion_config(mbstd_group_size=min(vbatch, 4), **kw)
                conv2d_gradfix.conv_math = 'bf16x6' if tag == 'conv_math_bf16x6' else math_before
                vstep = TrainingStep(device, cfg=vcfg, num_gpus=1, rank=0, batch_size=vbatch, batch_gpu=vbatch)
                vdata = data if (vbatch, vres) == (args.batch_gpu, 256) else SyntheticFullBodyBatch(vbatch, device, seed=rank, res=vres)
                for _ in range(2):
                    vstep.run(vdata)
                torch.cuda.synchronize()
                tv = time.perf_counter()
                for _ in range(16):
                    vstep.run(vdata)
                torch.cuda.synchronize()
                tv = time.perf_counter() - tv
                out['also_measured'][tag] = {'value': round(16 * vbatch / tv, 3), 'unit': 'images/sec', 'ms_per_step': round(1000 * tv / 16, 2),
                                             'steps': 16, 'warmup': 2, 'batch': vbatch, 'resolution': vres,
                                             'precision': 'fp32-equivalent, as the headline' if tag == 'conv_math_bf16x6' else 'REDUCED relative to the headline', 'note': note}
                conv2d_gradfix.conv_math = math_before
                del vstep, vdata
                torch.cuda.empty_cache()
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
