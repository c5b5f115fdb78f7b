"""Deterministic synthetic weights and inputs (no datasets or checkpoints exist offline).

Weights are generated per parameter name from a seeded CPU generator, with the same fan-based scale as the
reference's initialisers (xavier-uniform for the transformer / propagation matrices, kaiming-normal for the
convolutions), so the SAME state_dict can be rebuilt anywhere (build container, GPU box) and loaded into
the reference model, the oracle and this package's module alike.
"""
import hashlib
import math

import torch

CONFIGS = {
    # name: (constructor kwargs, forward kwargs) -- the reference's canonical flag sets (scripts/*.sh)
    'gmflow_s1': (dict(num_scales=1, upsample_factor=8, reg_refine=False, task='flow'),
                  dict(attn_type='swin', attn_splits_list=[2], corr_radius_list=[-1], prop_radius_list=[-1],
                       task='flow')),
    'gmflow_s2_rr6': (dict(num_scales=2, upsample_factor=4, reg_refine=True, task='flow'),
                      dict(attn_type='swin', attn_splits_list=[2, 8], corr_radius_list=[-1, 4],
                           prop_radius_list=[-1, 1], num_reg_refine=6, task='flow')),
    'gmstereo_s2_rr3': (dict(num_scales=2, upsample_factor=4, reg_refine=True, task='stereo'),
                        dict(attn_type='self_swin2d_cross_swin1d', attn_splits_list=[2, 8],
                             corr_radius_list=[-1, 4], prop_radius_list=[-1, 1], num_reg_refine=3, task='stereo')),
    'gmstereo_s1': (dict(num_scales=1, upsample_factor=8, reg_refine=False, task='stereo'),
                    dict(attn_type='self_swin2d_cross_1d', attn_splits_list=[2], corr_radius_list=[-1],
                         prop_radius_list=[-1], task='stereo')),
    'gmdepth_s1': (dict(num_scales=1, upsample_factor=8, reg_refine=False, task='depth'),
                   dict(attn_type='swin', attn_splits_list=[2], prop_radius_list=[-1], task='depth',
                        min_depth=0.1, max_depth=2.0, num_depth_candidates=64)),
    'gmdepth_s1_rr1': (dict(num_scales=1, upsample_factor=8, reg_refine=True, task='depth'),
                       dict(attn_type='swin', attn_splits_list=[2], prop_radius_list=[-1], task='depth',
                            min_depth=0.1, max_depth=2.0, num_depth_candidates=64, num_reg_refine=1)),
}

# ScanNet demo intrinsics (demo/depth-scannet/intrinsic/intrinsic_depth.txt of the reference), 640x480
SCANNET_K = ((577.590698, 0.0, 318.905426), (0.0, 578.729797, 242.683609), (0.0, 0.0, 1.0))


def _gen(name, seed):
    digest = hashlib.sha256(f'{seed}:{name}'.encode()).digest()
    g = torch.Generator(device='cpu')
    g.manual_seed(int.from_bytes(digest[:7], 'little'))
    return g


def synth_state_dict(shapes, seed=326, refine_gain=1.0, feature_gain=1.0):
    """shapes: {name: torch.Size}.  Returns {name: fp32 tensor} with reference-like init statistics.

    ``feature_gain`` scales the encoder's output projection and the Transformer's LayerNorm affines.  At 1.0
    (reference-like random init) the matching logits reach +-230 and the model is a chaotic matcher whose fp32
    evaluation differs from an fp64 one by 3e-3 .. 7e-3 px (one scale) or tens of pixels (two scales + refinement).
    ``CONDITIONED`` (0.25, with ``refine_gain`` 0.02) keeps every softmax soft: the fp32 reference then agrees with
    fp64 to ~1e-5 px on all five BASELINE configs, so the north star's ABSOLUTE 1e-3 px gate is meaningful there.

    ``refine_gain`` scales the last convolution of the refinement flow head.  With random weights the
    GRU refinement loop is chaotic (the reference disagrees with ITSELF by tens of pixels between two CPU
    thread counts); a small gain keeps end-to-end parity tests of the refine configs meaningful.
    """
    out = {}
    for name, shape in shapes.items():
        g = _gen(name, seed)
        shape = tuple(shape)
        if name.endswith('norm1.weight') or name.endswith('norm2.weight'):
            t = torch.ones(shape) + 0.02 * torch.randn(shape, generator=g)
        elif len(shape) == 1:                                    # biases
            t = 0.02 * torch.randn(shape, generator=g)
        elif len(shape) == 2:                                    # Linear: xavier uniform
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        else:                                                    # conv: kaiming normal, fan_out
            fan_out = shape[0] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
        if name.startswith('refine.flow_head.conv2'):
            t = t * refine_gain
        if feature_gain != 1.0 and (name.startswith('backbone.conv2.') or
                                    (name.startswith('transformer.') and '.norm' in name)):
            t = t * feature_gain
        out[name] = t.float()
    return out


CONDITIONED = dict(feature_gain=0.25, refine_gain=0.02)


def synth_images(batch, height, width, seed=1000, kind='shift', normalized=False, blur=5):
    """A seeded image pair ``[B,3,H,W]`` x2.

    kind='noise': independent uniform noise; kind='shift': crops of one box-blurred (``blur`` x ``blur``)
    noise canvas displaced by (+6, -4) px (a pair with real correspondences).  ``normalized``: ImageNet-normalised
    scale (stereo / depth inputs) instead of 0..255 (flow inputs).
    """
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    if kind == 'noise':
        a = torch.rand(batch, 3, height, width, generator=g)
        b = torch.rand(batch, 3, height, width, generator=g)
    else:
        pad = 16
        canvas = torch.rand(batch, 3, height + 2 * pad, width + 2 * pad, generator=g)
        canvas = torch.nn.functional.avg_pool2d(canvas, blur, stride=1, padding=blur // 2)
        canvas = (canvas - canvas.amin()) / (canvas.amax() - canvas.amin())
        a = canvas[:, :, pad:pad + height, pad:pad + width]
        b = canvas[:, :, pad - 4:pad - 4 + height, pad + 6:pad + 6 + width]
    if normalized:
        mean = torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1)
        std = torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1)
        return ((a - mean) / std).contiguous(), ((b - mean) / std).contiguous()
    return (a * 255.0).contiguous(), (b * 255.0).contiguous()


def synth_camera(batch, height, width):
    """Intrinsics ``[B,3,3]`` (ScanNet, rescaled) and a small relative pose ``[B,4,4]``."""
    k = torch.tensor(SCANNET_K, dtype=torch.float32)
    k[0] *= width / 640.0
    k[1] *= height / 480.0
    ang = 0.05
    rot = torch.tensor([[math.cos(ang), 0.0, math.sin(ang)], [0.0, 1.0, 0.0], [-math.sin(ang), 0.0, math.cos(ang)]])
    pose = torch.eye(4)
    pose[:3, :3] = rot
    pose[:3, 3] = torch.tensor([0.12, -0.03, 0.05])
    return k[None].repeat(batch, 1, 1).contiguous(), pose[None].repeat(batch, 1, 1).contiguous()
