"""BASELINE config 5: 16-bit activation storage end to end (bf16 here; fp16 is the same code with the other operand type).
Generator (``synthesis_kwargs.act_dtype``) and discriminator (``half_dtype`` + ``num_fp16_res``) against the oracle run IN THE
SAME STORAGE TYPE: ``oracle/ref_ops.STORAGE`` rounds every tensor an operator hands on, and the convolution weights, to
bf16 while products and sums stay fp32 -- the arithmetic of the HIP path (one matrix-core product per multiply-add, fp32
accumulation and epilogues, one rounding per stored tensor).

Tolerance, stated: two evaluations in bf16 storage differ where a sum lands next to a rounding boundary (summation order),
i.e. by one unit of 2^-8 relative in a few elements per tensor, and the differences propagate through ~40 layers.  Forward
tensors are held to 4e-2 of their largest value and to 1.5e-2 rms; parameter gradients to cosine >= 0.995 and norm within
10 % (the gradient of the first layer has crossed every rounding twice).  (The fp32 path holds 1e-4 / 1e-3 on the same quantities: tests/test_models_gpu.py.)  The generator in 16-bit storage is
this package's extension (the reference's generator is fp32 only, networks.py:5747-5748): parity UNPINNED, oracle-only; the
discriminator's fp16 blocks are the reference's and are pinned by its fixture in tests/test_fullwidth.py."""

import pytest
import torch

from oracle import param_fill as PF

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


def _close(a, b, what, tol_max=4e-2, tol_rms=1.5e-2):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= tol_max * scale, (what, float((a - b).abs().max()) / scale)
    assert float((a - b).square().mean().sqrt()) <= tol_rms * float(b.square().mean().sqrt()) + 1e-12, what


def _grad_close(a, b, what):
    a, b = a.detach().float().cpu().flatten(), b.detach().float().cpu().flatten()
    cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
    assert cos >= 0.995, (what, cos)
    assert abs(float(a.norm()) / float(b.norm()) - 1) <= 0.10, (what, float(a.norm()), float(b.norm()))


@pytest.fixture
def bf16_oracle():
    from oracle import ref_ops as R
    R.STORAGE = BF16
    yield R
    R.STORAGE = None


def test_discriminator_in_bf16_storage(bf16_oracle):
    from oracle import ref_networks as RN
    from training import networks
    kw = dict(PF.D_KWARGS, num_fp16_res=6, half_dtype='bfloat16')
    D = PF.fill_module(networks.Discriminator(**kw)).train().requires_grad_(True)
    assert all(getattr(D, f'b{r}').use_fp16 and getattr(D, f'b{r}').half_dtype == BF16 for r in (256, 128, 64, 32, 16, 8))
    params = dict(D.named_parameters())
    sd = {k: v.detach().clone().requires_grad_(k in params) for k, v in list(D.named_parameters()) + list(D.named_buffers())}
    c = torch.tanh(PF.make_inputs(n=4, seed=1)['style_input'].mean(dim=[2, 3]).repeat(1, 13)[:, :512])
    x = PF.make_inputs(n=4, seed=1)['real_img']
    want = RN.discriminator(sd, x, c)
    keys = ['b256.conv0.weight', 'b64.conv1.weight', 'b16.skip.weight', 'b8.conv0.weight', 'b4.fc.weight']
    gwant = torch.autograd.grad(torch.nn.functional.softplus(-want).mean(), [sd[k] for k in keys])
    D = D.cuda()
    got = D(x.cuda(), c.cuda())
    assert got.dtype == torch.float32
    _close(got, want, 'logits')
    torch.nn.functional.softplus(-got).mean().backward()
    for k, g in zip(keys, gwant):
        assert dict(D.named_parameters())[k].grad.dtype == torch.float32
        _grad_close(dict(D.named_parameters())[k].grad, g, k)


def test_generator_in_bf16_storage(bf16_oracle):
    from oracle import ref_networks as RN
    from training import networks
    kw = dict(PF.G_KWARGS, synthesis_kwargs=dict(PF.G_KWARGS['synthesis_kwargs'], act_dtype='bfloat16'))
    G = PF.fill_module(networks.GeneratorFull(**kw)).train().requires_grad_(True)
    assert G.synthesis.act_dtype == BF16 and G.synthesis.b64.use_fp16 and G.synthesis.b64.half_dtype == BF16
    params = dict(G.named_parameters())
    sd = {k: v.detach().clone().requires_grad_(k in params) for k, v in list(G.named_parameters()) + list(G.named_buffers())}
    inp = PF.make_inputs(n=2, seed=0)
    args = (inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
            inp['denorm_upper_mask'], inp['denorm_lower_mask'])
    img, fin, par = RN.generator_full(sd, *args, img_resolution=256, conv_clamp=256, mapping_layers=1, noise_mode='const')
    keys = ['synthesis.b64.conv0.weight', 'synthesis.b256.torgb.weight', 'synthesis.spade_b128_2.spade0.conv_gamma.weight',
            'synthesis.texture_b256.conv1.weight', 'const_encoding.model.3.weight', 'synthesis.b128.merge_conv.weight']
    probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
    gwant = torch.autograd.grad(probe, [sd[k] for k in keys])
    G = G.cuda()
    gi, gf, gp = G(*[a.cuda() for a in args], noise_mode='const')
    assert gi.dtype == gf.dtype == torch.float32                          # images leave the network in fp32 as in the reference
    _close(gi, img, 'img'); _close(gf, fin, 'finetune_img'); _close(gp, par, 'pred_parsing')
    ((gi * inp['real_img'].cuda()).mean() + gf.square().mean() + 0.1 * gp.float().abs().mean()).backward()
    got = dict(G.named_parameters())
    for k, g in zip(keys, gwant):
        assert got[k].grad is not None and got[k].grad.dtype == torch.float32
        _grad_close(got[k].grad, g, k)


def test_intermediate_activations_really_are_16_bit():
    """The point of the configuration: what travels between layers is 2 bytes per element."""
    from training import networks
    kw = dict(PF.G_KWARGS, synthesis_kwargs=dict(PF.G_KWARGS['synthesis_kwargs'], act_dtype='bfloat16'))
    G = PF.fill_module(networks.GeneratorFull(**kw)).cuda().eval().requires_grad_(False)
    seen = {}
    def note(name):
        def hook(mod, inputs, out):
            seen[name] = (out[0] if isinstance(out, tuple) else out).dtype       # returns None: the output is left alone
        return hook
    hooks = [m.register_forward_hook(note(name)) for name, m in G.named_modules() if name in ('synthesis.b64.conv1', 'synthesis.spade_b128_1', 'const_encoding.model.2', 'style_encoding.feat_enc.1')]
    inp = {k: v.cuda() for k, v in PF.make_inputs(n=1, seed=0).items()}
    with torch.no_grad():
        G(inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
          inp['denorm_upper_mask'], inp['denorm_lower_mask'], noise_mode='const')
    for h in hooks:
        h.remove()
    assert len(seen) == 4 and all(d == BF16 for d in seen.values()), seen
