"""BASELINE config 5 at ITS OWN workload: 512x320 (tensor 512x512) training, bf16 activation storage with fp32
demodulation / accumulation / statistics (missing 2: bf16 storage was tested at 256 only, the 512 model in fp32
only).  Three parts:

* the resolution-generalised ``GeneratorFull`` at 512 in TRAINING mode with ``act_dtype='bfloat16'`` against the oracle run in
  the same storage type (``oracle/ref_ops.STORAGE``), forward tensors and parameter gradients, at ``channel_base=2048``
  (encoders and SPADE widths stay full) and batch 1 so the CPU oracle finishes in seconds;
* the ``Discriminator`` at 512 with every block b512..b8 in bf16 storage, same yardstick, batch 2;
* one full-size step of the configuration itself (``cfg=fashion`` widths, batch 8): finite, every parameter moves.

Tolerances: those of tests/test_storage16_gpu.py (two bf16 evaluation orders differ by a unit of 2^-8 in a few elements per
tensor and the differences propagate through ~45 layers): forward 4e-2 of the largest value / 1.5e-2 rms, parameter gradients
cosine >= 0.995 and norm within 10 %.  The 512 model is this package's generalisation (the reference ships no 512 class,
SURVEY F9) and the bf16 generator its extension: parity UNPINNED, oracle-only."""

import pytest
import torch

from oracle import param_fill as PF
from test_storage16_gpu import _close, _grad_close, bf16_oracle, BF16  # noqa: F401  (bf16_oracle is a fixture)

pytestmark = pytest.mark.gpu

G512_BF16 = dict(z_dim=0, c_dim=512, w_dim=512, img_resolution=512, img_channels=3, mapping_kwargs=dict(num_layers=1),
                 synthesis_kwargs=dict(channel_base=2048, channel_max=512, conv_clamp=256, act_dtype='bfloat16'))
D512_BF16 = dict(c_dim=512, img_resolution=512, img_channels=3, channel_base=2048, channel_max=512, conv_clamp=256,
                 num_fp16_res=7, half_dtype='bfloat16', epilogue_kwargs=dict(mbstd_group_size=2))


@pytest.mark.timeout(900)
def test_generator_512_training_mode_in_bf16_storage(bf16_oracle):
    from oracle import ref_networks as RN
    from training import networks
    G = PF.fill_module(networks.GeneratorFull(**G512_BF16)).train().requires_grad_(True)
    assert G.synthesis.act_dtype == BF16 and G.synthesis.b512.use_fp16 and G.synthesis.b512.half_dtype == BF16
    params = dict(G.named_parameters())
    sd = {k: v.detach().clone().requires_grad_(k in params) for k, v in list(G.named_parameters()) + list(G.named_buffers())}
    inp = PF.make_inputs(n=1, seed=0, res=512)
    args = (inp['gen_z'], inp['style_input'], inp['retain'], inp['pose'], inp['denorm_upper_input'], inp['denorm_lower_input'],
            inp['denorm_upper_mask'], inp['denorm_lower_mask'])
    img, fin, par = RN.generator_full(sd, *args, img_resolution=512, conv_clamp=256, mapping_layers=1, noise_mode='const')
    keys = ['synthesis.b512.conv0.weight', 'synthesis.b512.torgb.weight', 'synthesis.spade_b256_2.spade0.conv_gamma.weight',
            'synthesis.texture_b512.conv1.weight', 'const_encoding.model.7.weight', 'synthesis.b64.merge_conv.weight',
            'synthesis.b128.conv1.weight']
    probe = (img * inp['real_img']).mean() + fin.square().mean() + 0.1 * par.abs().mean()
    gwant = torch.autograd.grad(probe, [sd[k] for k in keys])
    # yardstick for the deepest gradients: how far bf16 storage moves the ORACLE's own gradient from its fp32 value
    bf16_oracle.STORAGE = None
    img32, fin32, par32 = RN.generator_full(sd, *args, img_resolution=512, conv_clamp=256, mapping_layers=1, noise_mode='const')
    g32 = torch.autograd.grad((img32 * inp['real_img']).mean() + fin32.square().mean() + 0.1 * par32.abs().mean(), [sd[k] for k in keys])
    bf16_oracle.STORAGE = BF16
    G = G.cuda()
    gi, gf, gp = G(*[a.cuda() for a in args], noise_mode='const')
    assert gi.shape == (1, 3, 512, 512) and gi.dtype == gf.dtype == torch.float32
    _close(gi, img, 'img'); _close(gf, fin, 'finetune_img'); _close(gp, par, 'pred_parsing')
    ((gi * inp['real_img'].cuda()).mean() + gf.square().mean() + 0.1 * gp.float().abs().mean()).backward()
    got = dict(G.named_parameters())
    for k, g, g_fp32 in zip(keys, gwant, g32):
        assert got[k].grad is not None and got[k].grad.dtype == torch.float32
        a, b, c = got[k].grad.detach().float().cpu().flatten(), g.flatten(), g_fp32.flatten()
        cos_rounding = float(torch.dot(b, c) / (b.norm() * c.norm()))          # bf16 oracle against fp32 oracle
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        # 0.995 as at 256 -- or, for a gradient that has crossed every rounding of the one-level-deeper 512 model twice (the pose
        # encoder's last stage: 0.994 measured), as close to the bf16 oracle as bf16 storage leaves that oracle to fp32.  That yardstick is
        # itself ONE realisation of the rounding noise it measures: round 5's few-channel pointwise kernels (fp32 FMA chains where the fp32
        # MFMA tiles summed in another order: 1e-7 on a layer's output) move which elements round the other way downstream and read
        # 0.99393 against 0.99411 for that key, 0.99420 with the old kernels -- hence 5e-4 of slack on the yardstick, none on the 0.99 floor
        assert cos >= min(0.995, cos_rounding - 5e-4), (k, cos, cos_rounding)
        assert cos >= 0.99, (k, cos)
        assert abs(float(a.norm()) / float(b.norm()) - 1) <= 0.10, (k, float(a.norm()), float(b.norm()))


@pytest.mark.timeout(900)
def test_discriminator_512_in_bf16_storage(bf16_oracle):
    from oracle import ref_networks as RN
    from training import networks
    D = PF.fill_module(networks.Discriminator(**D512_BF16)).train().requires_grad_(True)
    assert all(getattr(D, f'b{r}').use_fp16 and getattr(D, f'b{r}').half_dtype == BF16 for r in (512, 256, 128, 64, 32, 16, 8))
    params = dict(D.named_parameters())
    sd = {k: v.detach().clone().requires_grad_(k in params) for k, v in list(D.named_parameters()) + list(D.named_buffers())}
    inp = PF.make_inputs(n=2, seed=1, res=512)
    c = torch.tanh(inp['style_input'].mean(dim=[2, 3]).repeat(1, 13)[:, :512])
    x = inp['real_img']
    want = RN.discriminator(sd, x, c, img_resolution=512)
    keys = ['b512.conv0.weight', 'b512.conv1.weight', 'b256.skip.weight', 'b64.conv1.weight', 'b8.conv0.weight', 'b4.fc.weight']
    gwant = torch.autograd.grad(torch.nn.functional.softplus(-want).mean(), [sd[k] for k in keys])
    D = D.cuda()
    got = D(x.cuda(), c.cuda())
    assert got.dtype == torch.float32
    _close(got, want, 'logits')
    torch.nn.functional.softplus(-got).mean().backward()
    for k, g in zip(keys, gwant):
        _grad_close(dict(D.named_parameters())[k].grad, g, k)


@pytest.mark.timeout(900)
def test_one_config5_training_iteration_full_size():
    """cfg=fashion widths, 512x512 tensors, batch 8, bf16 storage: the step BASELINE config 5 runs on each of its GPUs
    (bench.py --storage bf16 --train-res 512 --batch-gpu 8)."""
    from training.training_loop_wo_flow_fullbody import TrainingStep, SyntheticFullBodyBatch, fashion_config
    dev = torch.device('cuda', 0)
    cfg = fashion_config(img_resolution=512, act_dtype='bfloat16')
    step = TrainingStep(dev, cfg=cfg, batch_size=8, batch_gpu=8)
    assert step.G.synthesis.act_dtype == BF16 and step.D.b512.half_dtype == BF16
    data = SyntheticFullBodyBatch(8, dev, seed=0, res=512)
    g0 = [p.detach().clone() for p in step.G.parameters()]
    d0 = [p.detach().clone() for p in step.D.parameters()]
    step.run(data)            # iteration 0: Gmain, Greg, Dmain, Dreg (R1 through the bf16 blocks)
    step.run(data)
    torch.cuda.synchronize()
    moved_g = sum(int(not torch.equal(a, b)) for a, b in zip(g0, step.G.parameters()))
    moved_d = sum(int(not torch.equal(a, b)) for a, b in zip(d0, step.D.parameters()))
    assert moved_d == len(d0)
    assert moved_g >= len(g0) - 6            # b4.const and the texture block's unused parameters never get gradients
    assert all(p.dtype == torch.float32 for p in step.G.parameters())         # fp32 masters
    assert all(torch.isfinite(p).all() for p in step.G.parameters()) and all(torch.isfinite(p).all() for p in step.D.parameters())
