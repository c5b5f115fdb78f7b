"""CPU oracle of the whole UniMatch forward (eval mode).  TEST INFRASTRUCTURE ONLY.

A functional restatement of ``UniMatch.forward`` (/root/reference/unimatch/unimatch.py:95-367) over a
plain ``state_dict``: the hot path goes through ``oracle.hotpath``; the parts that are *outside* the
hot path (CNN encoder, convex upsampler, refinement convolutions) are restated with stock
``torch.nn.functional`` ops on the CPU.  Used (a) to pin the product's end-to-end output on seeded
inputs, (b) as ``bench.py``'s ``cpu_baseline`` ("port"), (c) in float64 as the ground truth beside which
the reference's own fp32 noise floor is reported.  Parity status: PINNED by tests/golden (see
oracle/hotpath.py).
"""
import torch
import torch.nn.functional as F

from . import hotpath as hp

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


# ------------------------------------------------------------------ CNN encoder (backbone.py:39-133)
def _res_block(x, p, pre, stride):
    y = F.conv2d(x, p[pre + 'conv1.weight'], None, stride=stride, padding=1)
    y = F.relu(F.instance_norm(y))
    y = F.conv2d(y, p[pre + 'conv2.weight'], None, padding=1)
    y = F.relu(F.instance_norm(y))
    if pre + 'downsample.0.weight' in p:
        x = F.conv2d(x, p[pre + 'downsample.0.weight'], p[pre + 'downsample.0.bias'], stride=stride)
        x = F.instance_norm(x)
    return F.relu(x + y)


def cnn_encoder(img, p, num_scales, pre='backbone.'):
    """Returns feature maps from LOW to HIGH resolution (unimatch.py:64-79 reverses the list)."""
    x = F.conv2d(img, p[pre + 'conv1.weight'], None, stride=2, padding=3)
    x = F.relu(F.instance_norm(x))
    x = _res_block(x, p, pre + 'layer1.0.', 1)
    x = _res_block(x, p, pre + 'layer1.1.', 1)
    x = _res_block(x, p, pre + 'layer2.0.', 2)
    x = _res_block(x, p, pre + 'layer2.1.', 1)
    x = _res_block(x, p, pre + 'layer3.0.', 2 if num_scales == 1 else 1)
    x = _res_block(x, p, pre + 'layer3.1.', 1)
    x = F.conv2d(x, p[pre + 'conv2.weight'], p[pre + 'conv2.bias'])
    if num_scales == 1:
        return [x]
    wt = p[pre + 'trident_conv.weight']                       # weight-shared strided branches
    outs = [F.conv2d(x, wt, None, stride=2 ** i, padding=1) for i in range(num_scales)]
    return outs[::-1]


# ------------------------------------------------------------------ small geometry helpers
def warp(feature, flow):
    """Bilinear warp, zeros outside, align_corners (geometry.py:41-72)."""
    b, c, h, w = feature.shape
    pos = hp.pixel_grid(h, w, feature.dtype)[None] + flow
    gx = 2 * pos[:, 0] / (w - 1) - 1
    gy = 2 * pos[:, 1] / (h - 1) - 1
    return F.grid_sample(feature, torch.stack([gx, gy], -1), mode='bilinear',
                         padding_mode='zeros', align_corners=True)


def convex_upsample(flow, mask, factor, is_depth=False):
    """RAFT convex upsampling (utils.py:134-152)."""
    b, ch, h, w = flow.shape
    m = torch.softmax(mask.view(b, 1, 9, factor, factor, h, w), dim=2)
    nb = F.unfold((1 if is_depth else factor) * flow, [3, 3], padding=1).view(b, ch, 9, 1, 1, h, w)
    up = (m * nb).sum(2)                                       # [B, ch, f, f, h, w]
    return up.permute(0, 1, 4, 2, 5, 3).reshape(b, ch, factor * h, factor * w)


def rigid_flow(depth, intrinsics, pose):
    """Flow induced by depth + relative pose (geometry.py:99-195)."""
    b, h, w = depth.shape
    dt = depth.dtype
    grid = hp.pixel_grid(h, w, dt)
    homog = torch.cat([grid, torch.ones(1, h, w, dtype=dt)], 0).flatten(1)
    pts = (torch.inverse(intrinsics) @ homog) * depth.view(b, 1, -1)
    pts = pose[:, :3, :3] @ pts + pose[:, :3, 3:]
    proj = intrinsics @ pts
    z = proj[:, 2:].clamp(min=1e-3)
    return (proj[:, :2] / z).view(b, 2, h, w) - grid


# ------------------------------------------------------------------ refinement (reg_refine.py:78-119)
def update_block(net, inp, corr, flow, p, pre='refine.'):
    e = pre + 'encoder.'
    cor = F.relu(F.conv2d(corr, p[e + 'convc1.weight'], p[e + 'convc1.bias']))
    cor = F.relu(F.conv2d(cor, p[e + 'convc2.weight'], p[e + 'convc2.bias'], padding=1))
    flo = F.relu(F.conv2d(flow, p[e + 'convf1.weight'], p[e + 'convf1.bias'], padding=3))
    flo = F.relu(F.conv2d(flo, p[e + 'convf2.weight'], p[e + 'convf2.bias'], padding=1))
    mot = F.relu(F.conv2d(torch.cat([cor, flo], 1), p[e + 'conv.weight'], p[e + 'conv.bias'], padding=1))
    x = torch.cat([inp, mot, flow], 1)
    g = pre + 'gru.'
    for tag, pad in (('1', (0, 2)), ('2', (2, 0))):            # horizontal 1x5 then vertical 5x1
        hx = torch.cat([net, x], 1)
        z = torch.sigmoid(F.conv2d(hx, p[g + f'convz{tag}.weight'], p[g + f'convz{tag}.bias'], padding=pad))
        r = torch.sigmoid(F.conv2d(hx, p[g + f'convr{tag}.weight'], p[g + f'convr{tag}.bias'], padding=pad))
        q = torch.tanh(F.conv2d(torch.cat([r * net, x], 1), p[g + f'convq{tag}.weight'],
                                p[g + f'convq{tag}.bias'], padding=pad))
        net = (1 - z) * net + z * q
    fh = pre + 'flow_head.'
    delta = F.conv2d(F.relu(F.conv2d(net, p[fh + 'conv1.weight'], p[fh + 'conv1.bias'], padding=1)),
                     p[fh + 'conv2.weight'], p[fh + 'conv2.bias'], padding=1)
    mask = None
    if pre + 'mask.0.weight' in p:
        mask = F.conv2d(F.relu(F.conv2d(net, p[pre + 'mask.0.weight'], p[pre + 'mask.0.bias'], padding=1)),
                        p[pre + 'mask.2.weight'], p[pre + 'mask.2.bias'])
    return net, mask, delta


def upsampler_mask(flow2, feature, p):
    """The upsampler head's convex-combination logits (unimatch.py:56-58, 246-250): cat(flow, feature) -> 3x3 + ReLU -> 1x1."""
    x = torch.cat([flow2, feature], 1)
    x = F.relu(F.conv2d(x, p['upsampler.0.weight'], p['upsampler.0.bias'], padding=1))
    return F.conv2d(x, p['upsampler.2.weight'], p['upsampler.2.bias'])


def _upsampler(flow2, feature, p, factor, is_depth=False, taps=None):
    mask = upsampler_mask(flow2, feature, p)
    if taps is not None:
        taps['up_mask'] = mask
    return convex_upsample(flow2, mask, factor, is_depth=is_depth)


# ------------------------------------------------------------------ the forward
@torch.no_grad()
def unimatch_forward(p, img0, img1, *, num_scales=1, upsample_factor=8, reg_refine=False, task='flow',
                     attn_type='swin', attn_splits_list=(2,), corr_radius_list=(-1,),
                     prop_radius_list=(-1,), num_reg_refine=1, pred_bidir_flow=False,
                     intrinsics=None, pose=None, min_depth=1. / 0.5, max_depth=1. / 10,
                     num_depth_candidates=64, depth_from_argmax=False, pred_bidir_depth=False,
                     num_transformer_layers=6, taps=None):
    """Eval-mode forward; returns the final prediction tensor (the single entry of 'flow_preds').

    ``taps``: optional dict that receives named intermediates (used by kernel-level parity tests).
    """
    dt = img0.dtype
    p = {k: v.to(dt) for k, v in p.items()}
    if task == 'flow':
        mean = torch.tensor(IMAGENET_MEAN, dtype=dt).view(1, 3, 1, 1)
        std = torch.tensor(IMAGENET_STD, dtype=dt).view(1, 3, 1, 1)
        img0, img1 = (img0 / 255. - mean) / std, (img1 / 255. - mean) / std
    feats = cnn_encoder(torch.cat([img0, img1], 0), p, num_scales)
    nb = img0.shape[0]
    flow = None
    pred = None
    for s in range(num_scales):
        f0, f1 = feats[s][:nb], feats[s][nb:]
        if pred_bidir_flow and s > 0:
            f0, f1 = torch.cat([f0, f1], 0), torch.cat([f1, f0], 0)
        f0_ori, f1_ori = f0, f1
        if taps is not None:
            taps[f'backbone0_s{s}'], taps[f'backbone1_s{s}'] = f0, f1
        up = upsample_factor * 2 ** (num_scales - 1 - s)
        if task == 'depth':
            k_cur = intrinsics.to(dt).clone()
            k_cur[:, :2] = k_cur[:, :2] / up
        if s > 0:
            flow = F.interpolate(flow, scale_factor=2, mode='bilinear', align_corners=True) * 2
        if flow is not None:
            disp = torch.cat([-flow, torch.zeros_like(flow)], 1) if task == 'stereo' else flow
            f1 = warp(f1, disp)
            if taps is not None:
                taps[f'flow_up_s{s}'], taps[f'f1_warp_s{s}'] = flow, f1
        splits = attn_splits_list[s]
        prop_r = prop_radius_list[s]
        f0, f1 = hp.add_position(f0, f1, splits)
        tp = {k[len('transformer.'):]: v for k, v in p.items() if k.startswith('transformer.')}
        f0, f1 = hp.feature_transformer(f0, f1, tp, attn_type, splits, num_transformer_layers, taps=taps, tag=f'_s{s}')
        if taps is not None:
            taps[f'f0_s{s}'], taps[f'f1_s{s}'] = f0, f1
        if task == 'depth':
            cand = torch.linspace(min_depth, max_depth, num_depth_candidates).to(dt)
            fp = hp.depth_corr_softmax(f0, f1, k_cur, pose.to(dt), cand, depth_from_argmax, pred_bidir_depth)
        else:
            r = corr_radius_list[s]
            if r == -1:
                fp = (hp.global_corr_softmax_flow(f0, f1, pred_bidir_flow) if task == 'flow'
                      else hp.global_corr_softmax_stereo(f0, f1))
            else:
                fp = hp.local_corr_softmax(f0, f1, r, one_d=(task == 'stereo'))
        flow = fp if flow is None else flow + fp
        if task == 'stereo':
            flow = flow.clamp(min=0)
        if taps is not None:
            taps[f'flow_match_s{s}'] = flow
        if (pred_bidir_flow or pred_bidir_depth) and s == 0:
            f0 = torch.cat([f0, f1], 0)
        flow = hp.prop_local(f0, flow, p, prop_r) if prop_r > 0 else hp.prop_global(f0, flow, p)
        if taps is not None:
            taps[f'flow_prop_s{s}'] = flow
        if s < num_scales - 1:
            continue
        if not reg_refine:
            if task == 'stereo':
                pad = torch.cat([-flow, torch.zeros_like(flow)], 1)
                pred = -_upsampler(pad, f0, p, upsample_factor, taps=taps)[:, :1]
            elif task == 'depth':
                pad = torch.cat([flow, torch.zeros_like(flow)], 1)
                pred = _upsampler(pad, f0, p, upsample_factor, is_depth=True, taps=taps).clamp(min=min_depth, max=max_depth)[:, :1]
            else:
                pred = _upsampler(flow, f0, p, upsample_factor, taps=taps)
            continue
        pose_r = pose
        for it in range(num_reg_refine):
            if task == 'stereo':
                disp = torch.cat([-flow, torch.zeros_like(flow)], 1)
            elif task == 'depth':
                if pred_bidir_depth and it == 0:
                    k_cur = k_cur.repeat(2, 1, 1)
                    pose_r = torch.cat([pose, torch.inverse(pose)], 0)
                    f0_ori, f1_ori = torch.cat([f0_ori, f1_ori], 0), torch.cat([f1_ori, f0_ori], 0)
                disp = rigid_flow(1. / flow.squeeze(1), k_cur, pose_r.to(dt))
            else:
                disp = flow
            corr = hp.local_corr_with_flow(f0_ori, f1_ori, disp, 4)
            if taps is not None:
                taps[f'cost_it{it}'] = corr
            proj = F.conv2d(f0, p['refine_proj.weight'], p['refine_proj.bias'])
            net, inp = torch.tanh(proj[:, :128]), torch.relu(proj[:, 128:])
            net, mask, delta = update_block(net, inp, corr, flow, p)
            if taps is not None and it == num_reg_refine - 1 and mask is not None:
                taps['mask_last'] = mask
            if task == 'depth':
                flow = (flow - delta).clamp(min=min_depth, max=max_depth)
            else:
                flow = flow + delta
            if task == 'stereo':
                flow = flow.clamp(min=0)
            if taps is not None:
                taps[f'flow_it{it}'] = flow
            if it == num_reg_refine - 1:
                if task == 'depth':
                    pad = torch.cat([flow, torch.zeros_like(flow)], 1)
                    pred = _upsampler(pad, f0, p, upsample_factor, is_depth=True).clamp(
                        min=min_depth, max=max_depth)[:, :1]
                else:
                    pred = convex_upsample(flow, mask, upsample_factor)
    if taps is not None:
        taps['pred_raw'] = pred
    if task == 'stereo':
        pred = pred.squeeze(1)
    if task == 'depth':
        pred = 1. / pred.squeeze(1)
    return pred
