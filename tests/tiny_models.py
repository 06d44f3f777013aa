"""Stand-in generator / discriminator made of stock PyTorch ops, with the call signatures the loss and the
training step use. CPU-only test doubles for exercising the data-parallel orchestration under gloo."""

import torch
import torch.nn as nn
import torch.nn.functional as F


class TinyMapping(nn.Module):
    def __init__(self, c_dim=8, w_dim=8, num_ws=4):
        super().__init__()
        self.fc = nn.Linear(c_dim, w_dim)
        self.num_ws = num_ws
        self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, skip_w_avg_update=False):
        x = self.fc(c)
        if self.training and not skip_w_avg_update:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, 0.995))
        return x.unsqueeze(1).repeat(1, self.num_ws, 1)


class TinyConstEnc(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(6, 4, 3, padding=1)

    def forward(self, pose):
        return self.conv(pose)


class TinyStyleEnc(nn.Module):
    def __init__(self, c_dim=8):
        super().__init__()
        self.conv = nn.Conv2d(42, c_dim, 1)
        self.feat = nn.Conv2d(3, 2, 3, padding=1)

    def forward(self, x, retain):
        return self.conv(x).mean(dim=[2, 3]), [self.feat(retain)]


class TinySynthesis(nn.Module):
    def __init__(self, w_dim=8):
        super().__init__()
        self.affine = nn.Linear(w_dim, 4)
        self.conv = nn.Conv2d(4 + 2, 8, 3, padding=1)
        self.rgb = nn.Conv2d(8, 3, 1)
        self.fine = nn.Conv2d(8 + 6, 3, 3, padding=1)
        self.parse = nn.Conv2d(8, 6, 1)
        self.unused = nn.Parameter(torch.zeros([3]))      # never touched: exercises find_unused_parameters

    def forward(self, ws, pose_feat, cat_feats, du_in, dl_in, du_mask, dl_mask):
        s = self.affine(ws[:, 0])
        x = pose_feat * s[:, :, None, None]
        x = F.leaky_relu(self.conv(torch.cat([x, cat_feats[str(x.shape[2])]], dim=1)), 0.2)
        img = self.rgb(x)
        fin = self.fine(torch.cat([x, du_in * du_mask, dl_in * dl_mask], dim=1)) + img
        return img, fin, self.parse(x)


class TinyG(nn.Module):
    def __init__(self, z_dim=0, c_dim=8, w_dim=8, **_unused):
        super().__init__()
        self.z_dim = z_dim
        self.mapping = TinyMapping(c_dim, w_dim)
        self.synthesis = TinySynthesis(w_dim)
        self.const_encoding = TinyConstEnc()
        self.style_encoding = TinyStyleEnc(c_dim)


class TinyD(nn.Module):
    def __init__(self, c_dim=8, **_unused):
        super().__init__()
        self.conv = nn.Conv2d(3, 8, 3, padding=1)
        self.fc = nn.Linear(8, c_dim)

    def forward(self, img, c):
        x = F.leaky_relu(self.conv(img), 0.2).mean(dim=[2, 3])
        return (self.fc(x) * c).sum(dim=1, keepdim=True)
