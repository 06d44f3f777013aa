"""TEST INFRASTRUCTURE: CPU restatement of the reference's VGG-19 perceptual term with stock PyTorch ops.

Follows training/loss_wo_flow_fullbody.py:259-310 (``VGGLoss`` over ``VGG19_Feature``: relu{1..5}_1 of torchvision's
VGG-19 'E' configuration, listed at :385).  PARITY UNPINNED for this term: the reference's class cannot be instantiated
here (it loads ./checkpoints/vgg19-dcbb9e9d.pth in its constructor, and torchvision is absent), so this restatement is
anchored on the published layer table only."""

import torch
import torch.nn.functional as F

CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
TAP_LAYERS = (1, 6, 11, 20, 29)          # indices into torchvision's ``features`` whose outputs are returned (:293-302)
WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)


def features(x, state):
    """``state``: {'features.<i>.weight', 'features.<i>.bias'} as in torchvision's checkpoint."""
    out, idx = [], 0
    for v in CFG:
        if v == 'M':
            x = F.max_pool2d(x, kernel_size=2, stride=2)
            idx += 1
        else:
            x = F.conv2d(x, state[f'features.{idx}.weight'], state[f'features.{idx}.bias'], padding=1)
            idx += 1
            x = F.relu(x)
            if idx in TAP_LAYERS:
                out.append(x)
            idx += 1
        if len(out) == len(TAP_LAYERS):
            break
    return out


def vgg_loss(x, y, state):
    fx, fy = features(x, state), features(y, state)
    return sum(w * F.l1_loss(a, b.detach()) for w, a, b in zip(WEIGHTS, fx, fy))
