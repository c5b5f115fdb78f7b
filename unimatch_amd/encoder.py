"""CNN feature encoder.  On the GPU it runs channels-last on the library's convolution / InstanceNorm kernels
(``_forward_nhwc``: ``um_conv7_fwd`` stem, ``um_conv2d_fwd``, ``um_nhwc_instance_norm``); the stock ``nn.Conv2d`` modules below hold
the parameters and are the CPU path.

Architecture and parameter names follow /root/reference/unimatch/backbone.py:39-133 and
unimatch/trident_conv.py:10-90 so that reference checkpoints load unchanged
(``backbone.conv1.weight``, ``backbone.layer2.0.downsample.0.bias``, ``backbone.trident_conv.weight`` ...).
"""

import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)


class _Residual(nn.Module):
    """Two 3x3 convs with InstanceNorm + ReLU, projection shortcut when the shape changes."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1, self.conv2 = _conv3(cin, cout, stride), _conv3(cout, cout)
        self.norm1, self.norm2 = nn.InstanceNorm2d(cout), nn.InstanceNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.norm3 = nn.InstanceNorm2d(cout)
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride), self.norm3)

    def forward(self, x, ops=None):
        if ops is not None:            # fused InstanceNorm + ReLU (+ shortcut + ReLU) kernels
            y = ops.instance_norm(self.conv1(x), relu=True)
            sc = x if self.downsample is None else ops.instance_norm(self.downsample[0](x), relu=False)
            return ops.instance_norm(self.conv2(y), relu=True, shortcut=sc.contiguous())
        y = F.relu(self.norm1(self.conv1(x)))
        y = F.relu(self.norm2(self.conv2(y)))
        return F.relu((x if self.downsample is None else self.downsample(x)) + y)


    def forward_nhwc(self, act, ops, want_f32):
        """Channels-last path on the matrix cores: ``act = (planes, f32 | None, b, h, w, c)`` (operand planes of the block
        input).  Every normalisation kernel writes the next convolution's operand planes directly; the identity shortcut is
        read back from those planes (hi + lo), so no fp32 copy of a block's input exists; ``want_f32`` asks for an fp32 copy
        of the output as well."""
        planes, f32, b, h, w, c = act
        stride = self.conv1.stride[0]
        cout = self.conv1.out_channels
        t, ho, wo = ops.conv2d_nhwc((planes, b, h, w, c), self.conv1.weight, None, stride, (1, 1), stats=True)
        tp, _ = ops.nhwc_norm(t, b, ho * wo, relu=True, want_planes=True, conv_stats=ops.last_conv_stats)
        u, _, _ = ops.conv2d_nhwc((tp, b, ho, wo, cout), self.conv2.weight, None, 1, (1, 1), stats=True)
        ustats = ops.last_conv_stats
        sc = scp = None
        if self.downsample is None:
            sc, scp = (f32, None) if f32 is not None else (None, planes)
        else:
            proj = self.downsample[0]
            d, _, _ = ops.conv2d_nhwc((planes, b, h, w, c), proj.weight, proj.bias, stride, (0, 0), stats=True)
            _, sc = ops.nhwc_norm(d, b, ho * wo, relu=False, want_planes=False, want_f32=True, conv_stats=ops.last_conv_stats)
        op, of = ops.nhwc_norm(u, b, ho * wo, relu=True, shortcut=sc, shortcut_planes=scp, want_planes=True, want_f32=want_f32,
                               conv_stats=ustats)
        return op, of, b, ho, wo, cout


class _SharedStridedConv(nn.Module):
    """One 3x3 weight applied at strides 1, 2, 4... ("trident" multi-scale branches, no bias)."""

    def __init__(self, channels, num_branch):
        super().__init__()
        self.num_branch = num_branch
        self.weight = nn.Parameter(torch.empty(channels, channels, 3, 3))
        nn.init.kaiming_uniform_(self.weight, nonlinearity='relu')

    def forward(self, x):
        return [F.conv2d(x, self.weight, None, stride=2 ** i, padding=1) for i in range(self.num_branch)]

    def forward_nhwc(self, act, ops):
        planes, _, b, h, w, c = act
        outs = []
        for i in range(self.num_branch):
            o, ho, wo = ops.conv2d_nhwc((planes, b, h, w, c), self.weight, None, 2 ** i, (1, 1))
            outs.append(o.view(b, ho, wo, -1).permute(0, 3, 1, 2))
        return outs


class CNNEncoder(nn.Module):
    """1/8 features (one scale) or 1/4 + 1/8 features (two scales, weight-shared strided conv)."""

    def __init__(self, output_dim=128, num_output_scales=1):
        super().__init__()
        self.num_branch = num_output_scales
        d0, d1, d2 = 64, 96, 128
        self.conv1 = nn.Conv2d(3, d0, 7, stride=2, padding=3, bias=False)
        self.norm1 = nn.InstanceNorm2d(d0)
        self.layer1 = nn.Sequential(_Residual(d0, d0, 1), _Residual(d0, d0, 1))
        self.layer2 = nn.Sequential(_Residual(d0, d1, 2), _Residual(d1, d1, 1))
        self.layer3 = nn.Sequential(_Residual(d1, d2, 2 if num_output_scales == 1 else 1), _Residual(d2, d2, 1))
        self.conv2 = nn.Conv2d(d2, output_dim, 1)
        if num_output_scales > 1:
            if num_output_scales > 4:
                raise ValueError('at most 4 output scales')
            self.trident_conv = _SharedStridedConv(output_dim, num_output_scales)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    @staticmethod
    def takes_raw_images(ops, x):
        """True when the channels-last path will run: it folds the input normalisation into the stem's image packing."""
        return ops is not None and getattr(ops, 'fused_conv', False) and x.is_cuda

    def forward(self, x, ops=None, input_norm=None):
        """``ops``: a backend offering ``instance_norm`` (HipOps) fuses the normalisation / activation tail of
        every convolution; ``None`` keeps the stock PyTorch modules (CPU tests).  ``input_norm = (mean3, std3)``: the
        images are raw 0..255 and still need ``(x / 255 - mean) / std`` (only passed when ``takes_raw_images``)."""
        if self.takes_raw_images(ops, x):
            return self._forward_nhwc(x, ops, input_norm)
        assert input_norm is None
        if ops is not None and getattr(ops, 'fused_tail', False) and x.is_cuda:
            x = ops.instance_norm(self.conv1(x), relu=True)
            for layer in (self.layer1, self.layer2, self.layer3):
                for block in layer:
                    x = block(x, ops)
            x = self.conv2(x)
        else:
            x = F.relu(self.norm1(self.conv1(x)))
            x = self.conv2(self.layer3(self.layer2(self.layer1(x))))
        return self.trident_conv(x) if self.num_branch > 1 else [x]       # high -> low resolution


    shortcut_f32 = False

    def _forward_nhwc(self, x, ops, input_norm=None):
        """The whole encoder in channels-last layout on the library's convolution / normalisation kernels
        (``um_stem_conv_fwd``, ``um_conv2d_fwd``, ``um_nhwc_instance_norm``).  The returned maps are NCHW *views* of NHWC memory, so
        ``flatten(2).transpose(1, 2)`` downstream (token-major features) is free."""
        b = x.shape[0]
        y, h, w = ops.stem_conv(x.contiguous(), self.conv1.weight, input_norm, stats=True)    # fp32 NHWC [b*h*w, 64]
        c = y.shape[1]
        keep_f32 = self.shortcut_f32                            # A/B knob (tools/ab_bench.py --set): fp32 copies for the shortcuts
        planes, f32 = ops.nhwc_norm(y, b, h * w, relu=True, want_planes=True, want_f32=keep_f32, conv_stats=ops.last_conv_stats)
        act = (planes, f32, b, h, w, c)
        blocks = [blk for layer in (self.layer1, self.layer2, self.layer3) for blk in layer]
        for i, blk in enumerate(blocks):
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            act = blk.forward_nhwc(act, ops, want_f32=keep_f32 and nxt is not None and nxt.downsample is None)
        planes, _, b, h, w, c = act
        out, _, _ = ops.conv2d_nhwc((planes, b, h, w, c), self.conv2.weight, self.conv2.bias, 1, (0, 0))
        if self.num_branch == 1:
            return [out.view(b, h, w, -1).permute(0, 3, 1, 2)]
        tp, _ = ops.nhwc_norm(out, b, h * w, normalize=False, relu=False, want_planes=True)
        return self.trident_conv.forward_nhwc((tp, None, b, h, w, out.shape[1]), ops)
