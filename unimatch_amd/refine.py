"""RAFT-style regression refinement block and convex upsampling: parameter container + CPU path (on the GPU the block runs
channels-last on the library's convolutions, ``unimatch_amd/refine_nhwc.py``).

Parameter names follow /root/reference/unimatch/reg_refine.py (``refine.encoder.convc1.weight``,
``refine.gru.convz1.bias``, ``refine.flow_head.conv2.weight``, ``refine.mask.2.weight`` ...).  The block's
correlation input is produced by the HIP kernel ``um_local_corr_with_flow``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _MotionEncoder(nn.Module):
    def __init__(self, corr_channels, flow_channels):
        super().__init__()
        self.convc1 = nn.Conv2d(corr_channels, 256, 1)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(flow_channels, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(256, 128 - flow_channels, 3, padding=1)

    def forward(self, flow, corr):
        c = F.relu(self.convc2(F.relu(self.convc1(corr))))
        f = F.relu(self.convf2(F.relu(self.convf1(flow))))
        return torch.cat([F.relu(self.conv(torch.cat([c, f], 1))), flow], 1)


class _SepConvGRU(nn.Module):
    """GRU with separable 1x5 then 5x1 gates."""

    def __init__(self, hidden, inputs):
        super().__init__()
        cin = hidden + inputs
        for tag, ks, pad in (('1', (1, 5), (0, 2)), ('2', (5, 1), (2, 0))):
            for gate in 'zrq':
                setattr(self, f'conv{gate}{tag}', nn.Conv2d(cin, hidden, ks, padding=pad))

    def forward(self, h, x):
        for tag in ('1', '2'):
            hx = torch.cat([h, x], 1)
            z = torch.sigmoid(getattr(self, 'convz' + tag)(hx))
            r = torch.sigmoid(getattr(self, 'convr' + tag)(hx))
            q = torch.tanh(getattr(self, 'convq' + tag)(torch.cat([r * h, x], 1)))
            h = (1 - z) * h + z * q
        return h


class _FlowHead(nn.Module):
    def __init__(self, cin, hidden, cout):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, hidden, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden, cout, 3, padding=1)

    def forward(self, x):
        return self.conv2(F.relu(self.conv1(x)))


class BasicUpdateBlock(nn.Module):
    def __init__(self, corr_channels=81, hidden_dim=128, context_dim=128, downsample_factor=8, flow_dim=2,
                 bilinear_up=False):
        super().__init__()
        self.encoder = _MotionEncoder(corr_channels, flow_dim)
        self.gru = _SepConvGRU(hidden_dim, context_dim + hidden_dim)
        self.flow_head = _FlowHead(hidden_dim, 256, flow_dim)
        self.mask = None
        if not bilinear_up:
            self.mask = nn.Sequential(nn.Conv2d(hidden_dim, 256, 3, padding=1), nn.ReLU(inplace=True),
                                      nn.Conv2d(256, downsample_factor ** 2 * 9, 1))

    def forward(self, net, inp, corr, flow):
        net = self.gru(net, torch.cat([inp, self.encoder(flow, corr)], 1))
        return net, (self.mask(net) if self.mask is not None else None), self.flow_head(net)


def convex_upsample(flow, mask, factor, is_depth=False):
    """9-tap convex combination upsampling (reference: unimatch/utils.py:134-152)."""
    b, ch, h, w = flow.shape
    weights = torch.softmax(mask.view(b, 1, 9, factor, factor, h, w), dim=2)
    taps = F.unfold((1 if is_depth else factor) * flow, [3, 3], padding=1).view(b, ch, 9, 1, 1, h, w)
    up = (weights * taps).sum(2)
    return up.permute(0, 1, 4, 2, 5, 3).reshape(b, ch, factor * h, factor * w)
