"""Host logic of the loss on the CPU: one stacked discriminator pass must reproduce the reference's separate calls
(loss_wo_flow_fullbody.py:127-128, 214-215, 235) for every relation between the batch size and the minibatch-std group
size (networks.py:1007-1022) -- including batches smaller than the group, where stacking would change the groups."""

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from training import networks
from training.loss_wo_flow_fullbody import StyleGAN2Loss


class _MbstdD(nn.Module):
    """Stock-op discriminator with the package's MinibatchStdLayer (pure tensor algebra, runs on the CPU)."""
    def __init__(self, group_size):
        super().__init__()
        self.conv = nn.Conv2d(3, 4, 3, padding=1)
        self.mbstd = networks.MinibatchStdLayer(group_size=group_size)
        self.fc = nn.Linear(5, 6)

    def forward(self, img, c):
        x = self.mbstd(F.leaky_relu(self.conv(img), 0.2)).mean(dim=[2, 3])
        return (self.fc(x) * c).sum(dim=1, keepdim=True)


@pytest.mark.parametrize('group_size', [4, None])
@pytest.mark.parametrize('k', [2, 3])
@pytest.mark.parametrize('n', [1, 2, 3, 4, 8])
def test_run_D_multi_equals_separate_calls(n, k, group_size):
    torch.manual_seed(n * 10 + k)
    D = _MbstdD(group_size)
    loss = StyleGAN2Loss(device=torch.device('cpu'), G_mapping=None, G_synthesis=None, G_const_encoding=None, G_style_encoding=None,
                         D=D, contextual_weight=0, vgg_weight=0)
    imgs = [torch.randn([n, 3, 8, 8]) for _ in range(k)]
    cs = [torch.randn([n, 6]) for _ in range(k)]
    merged = loss.run_D_multi(imgs, cs, sync=True)
    for got, img, c in zip(merged, imgs, cs):
        want = loss.run_D(img, c, sync=True)
        assert got.shape == want.shape
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), (n, k, group_size)


def test_mbstd_groups_rule():
    loss = StyleGAN2Loss(device=torch.device('cpu'), G_mapping=None, G_synthesis=None, G_const_encoding=None, G_style_encoding=None,
                         D=_MbstdD(4), contextual_weight=0, vgg_weight=0)
    assert loss._mbstd_groups(16) == 4 and loss._mbstd_groups(4) == 1 and loss._mbstd_groups(8) == 2
    assert loss._mbstd_groups(2) is None and loss._mbstd_groups(3) is None and loss._mbstd_groups(6) is None
    loss.D = _MbstdD(None)
    assert loss._mbstd_groups(8) is None
    loss.D = nn.Conv2d(3, 3, 1)
    assert loss._mbstd_groups(5) == 5
