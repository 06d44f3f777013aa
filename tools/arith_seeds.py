"""VERDICT r3 "do this" 1a: is PASTA_MATH_F16X3 systematically further from the reference than the other two fp32-class
arithmetics, or was four-for-four on ONE seed luck?

Two comparisons -- the narrow generator (tests/test_models_gpu.py) and the full-width discriminator's Dreg phase
(tests/test_fullwidth.py) -- over SEEDS input seeds, each against the PINNED oracle (oracle/ref_networks.py, held to the
reference's own fixtures by tests/test_oracle_golden.py / test_fullwidth.py) evaluated in DOUBLE precision: what the reference's
algorithm computes, free of any one fp32 evaluation order.

    python tools/arith_seeds.py ref  [--seeds 6]      CPU: fp64 oracle -> profiles/r4_arith_seeds_ref.npz (sampled tensors)
    python tools/arith_seeds.py gpu  [--seeds 6]      MI355X: f16x3 / bf16x6 / f32 against it -> the table on stdout

The oracle fixes its working precision with the NAME torch.float32 in a few casts (as the reference does); for the fp64 run the
name is bound to the double type (the accommodation of oracle/make_golden_models_f64.py).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'pasta-gan_amd'), ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch

from oracle import param_fill as PF, make_golden_fullwidth as FW, make_golden_fullwidth_r1 as R1
from oracle.make_golden_models import GRAD_KEYS_G

REF = os.path.join(ROOT, 'profiles', 'r4_arith_seeds_ref.npz')
NS = 2048


def sample(t):
    flat = t.detach().reshape(-1)
    step = max(flat.numel() // NS, 1)
    return flat[::step][:NS].double().cpu().numpy().copy()


def g_args(inp):
    return (inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
            inp['denorm_upper_mask'], inp['denorm_lower_mask'])


def d_inputs(seed):
    inp = PF.make_inputs(n=R1.BATCH, seed=100 + seed)
    c = torch.tanh(inp['style_input'].mean(dim=[2, 3]).repeat(1, 13)[:, :512] * 8)
    return inp['real_img'], c


def run_ref(seeds):
    from oracle import ref_networks as RN
    from training import networks
    out = {}
    real_f32 = torch.float32
    def state(cls, kw, kind):
        m = PF.fill_module(cls(**kw), kind=kind)
        params = dict(m.named_parameters())
        sd = {k: v.detach().double().clone().requires_grad_(k in params) for k, v in list(m.named_parameters()) + list(m.named_buffers())}
        return sd, sorted(params)
    sdG, pG = state(networks.GeneratorFull, PF.G_KWARGS, 'wave')
    sdD, pD = state(networks.Discriminator, FW.D_KWARGS, 'normal')
    for s in range(seeds):
        inp = {k: (v.double() if v.is_floating_point() else v) for k, v in PF.make_inputs(n=2, seed=s).items()}
        torch.float32 = torch.float64
        try:
            img, fin, par = RN.generator_full(sdG, *g_args(inp), img_resolution=256, conv_clamp=256, mapping_layers=1, noise_mode='const')
            assert img.dtype == torch.float64
            probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
            grads = dict(zip(pG, torch.autograd.grad(probe, [sdG[k] for k in pG], allow_unused=True)))
        finally:
            torch.float32 = real_f32
        out[f'G{s}.img'] = sample(img)
        out[f'G{s}.probe'] = np.array([probe.item()])
        for k in GRAD_KEYS_G:
            out[f'G{s}.grad.{k}'] = sample(grads[k])
        out[f'G{s}.gradnorms'] = np.array([float(grads[k].norm()) if grads[k] is not None else -1.0 for k in pG])
        print('generator seed', s, 'probe', probe.item(), flush=True)
        x, c = d_inputs(s)
        torch.float32 = torch.float64
        try:
            logits, gx, pen, dg = R1.dreg_phase(lambda im: RN.discriminator(sdD, im, c.double()), x.double(), [sdD[k] for k in pD])
            assert logits.dtype == torch.float64
        finally:
            torch.float32 = real_f32
        dg = dict(zip(pD, dg))
        out[f'D{s}.logits'] = logits.detach().numpy().copy()
        out[f'D{s}.gx'] = sample(gx)
        out[f'D{s}.gx_l2'] = np.array([float(gx.detach().norm())])
        out[f'D{s}.pen'] = pen.detach().numpy().copy()
        for k in FW.GRAD_KEYS_D:
            if dg[k] is not None:
                out[f'D{s}.grad.{k}'] = sample(dg[k])
        out[f'D{s}.gradnorms'] = np.array([float(dg[k].norm()) if dg[k] is not None else -1.0 for k in pD])
        print('discriminator seed', s, 'logits', logits.detach().flatten().tolist(), 'penalty', pen.detach().tolist(), flush=True)
    np.savez_compressed(REF, **out)
    print('written', REF, len(out), 'arrays')


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def run_gpu(seeds):
    from training import networks
    from torch_utils.ops import conv2d_gradfix as cg
    ref = np.load(REF)
    modes = ['f16x3', 'bf16x6', 'f32']
    rows = []
    G = PF.fill_module(networks.GeneratorFull(**PF.G_KWARGS)).cuda().train().requires_grad_(True)
    D = PF.fill_module(networks.Discriminator(**FW.D_KWARGS), kind='normal').cuda().train().requires_grad_(True)
    pG, pD = sorted(dict(G.named_parameters())), sorted(dict(D.named_parameters()))
    sdG, sdD = dict(G.named_parameters()), dict(D.named_parameters())
    for s in range(seeds):
        for mode in modes:
            cg.conv_math = mode
            inp = {k: v.cuda() for k, v in PF.make_inputs(n=2, seed=s).items()}
            G.zero_grad(set_to_none=True)
            img, fin, par = G(*g_args(inp), noise_mode='const')
            probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
            probe.backward()
            e_img = rel(sample(img), ref[f'G{s}.img'])
            eg = {k: rel(sample(sdG[k].grad), ref[f'G{s}.grad.{k}']) for k in GRAD_KEYS_G}
            gn = np.array([float(sdG[k].grad.double().norm()) if sdG[k].grad is not None else -1.0 for k in pG])
            rn = ref[f'G{s}.gradnorms']
            ok = rn > 2e-6 * rn.max()
            e_gn = float(np.abs(gn[ok] / rn[ok] - 1).max())
            worst = max(eg, key=eg.get)
            rows.append(('G', s, mode, dict(img=e_img, grad_max=eg[worst], grad_median=float(np.median(list(eg.values()))),
                                            gradnorm_max=e_gn, probe=abs(probe.item() / float(ref[f'G{s}.probe'][0]) - 1)), worst))
            # Dreg phase
            x, c = (t.cuda() for t in d_inputs(s))
            x = x.detach().requires_grad_(True)
            D.zero_grad(set_to_none=True)
            logits = D(x, c)
            with cg.no_weight_gradients():
                gx, = torch.autograd.grad([logits.sum()], [x], create_graph=True)
            pen = gx.square().sum([1, 2, 3])
            ((logits * 0 + pen * (R1.R1_GAMMA / 2)).mean() * R1.GAIN).backward()
            ed = {k: rel(sample(sdD[k].grad), ref[f'D{s}.grad.{k}']) for k in FW.GRAD_KEYS_D if f'D{s}.grad.{k}' in ref}
            gn = np.array([float(sdD[k].grad.double().norm()) if sdD[k].grad is not None else -1.0 for k in pD])
            rn = ref[f'D{s}.gradnorms']
            ok = rn > 1e-6 * rn.max()
            worst = max(ed, key=ed.get)
            rows.append(('D', s, mode, dict(logits=rel(logits.detach().cpu().numpy(), ref[f'D{s}.logits']), gx_samples=rel(sample(gx), ref[f'D{s}.gx']),
                                            gx_l2=abs(float(gx.detach().double().norm()) / float(ref[f'D{s}.gx_l2'][0]) - 1),
                                            penalty=rel(pen.detach().cpu().numpy(), ref[f'D{s}.pen']), grad_max=ed[worst],
                                            grad_median=float(np.median(list(ed.values()))), gradnorm_max=float(np.abs(gn[ok] / rn[ok] - 1).max())), worst))
        cg.conv_math = 'default'
    for net in ('G', 'D'):
        keys = list(next(r for r in rows if r[0] == net)[3])
        print(f'\n== {"narrow GeneratorFull (probe backward)" if net == "G" else "full-width Discriminator, Dreg phase (every gradient a second derivative)"}:'
              f' max relative deviation from the oracle in fp64, per input seed ==')
        print(f'{"seed":>4} {"arithmetic":>10} ' + ' '.join(f'{k:>13}' for k in keys) + '  worst gradient key')
        for r in rows:
            if r[0] == net:
                print(f'{r[1]:>4} {r[2]:>10} ' + ' '.join(f'{r[3][k]:>13.3e}' for k in keys) + '  ' + r[4])
        print('-- over seeds: median (max) --')
        for mode in modes:
            sel = [r[3] for r in rows if r[0] == net and r[2] == mode]
            print(f'{"":>4} {mode:>10} ' + ' '.join(f'{np.median([q[k] for q in sel]):.1e}({max(q[k] for q in sel):.0e})'.rjust(13) for k in keys))
        # rank statistics: how often is each arithmetic the furthest of the three?
        for k in keys:
            worst_count = {m: 0 for m in modes}
            for s in range(seeds):
                trio = {r[2]: r[3][k] for r in rows if r[0] == net and r[1] == s}
                worst_count[max(trio, key=trio.get)] += 1
            print(f'   furthest of the three on {k:>13}: ' + ', '.join(f'{m} {worst_count[m]}/{seeds}' for m in modes))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('what', choices=['ref', 'gpu'])
    ap.add_argument('--seeds', type=int, default=6)
    a = ap.parse_args()
    (run_ref if a.what == 'ref' else run_gpu)(a.seeds)
