"""``grid_sample`` entry point kept for API compatibility (reference: grid_sample_gradfix.py:22-30).

Only ADA augmentation calls it (training/augment.py), which is outside the generator /
discriminator hot path; it forwards to PyTorch-ROCm's own bilinear sampler, whose double backward
is available on current PyTorch."""

import torch

enabled = False  # Kept for API compatibility.

def grid_sample(input, grid):
    return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)
