"""Drop-in ``UniMatch`` module whose global-matching hot path runs on hand-written gfx950 HIP kernels.

Boundary (what a caller sees) is the reference's own:
``UniMatch(num_scales, feature_channels, upsample_factor, num_head, ffn_dim_expansion,
num_transformer_layers, reg_refine, task)`` (/root/reference/unimatch/unimatch.py:17-26),
``forward(img0, img1, attn_type, attn_splits_list, corr_radius_list, prop_radius_list, num_reg_refine,
pred_bidir_flow, task, intrinsics, pose, min_depth, max_depth, num_depth_candidates, depth_from_argmax,
pred_bidir_depth)`` -> ``{'flow_preds': [tensor]}`` (unimatch.py:95-111, 365-367) and the reference's
``state_dict`` names, so ``evaluate_flow/stereo/depth`` can drive it unchanged.

What is different inside:
  * features travel token-major ``[N, h*w, C]`` through the whole hot path (no permute/copy per window split);
  * self / cross attention is decided by the layer's role, not by a device->host sync on the data
    (reference: ``(query - key).abs().max() < 1e-6``, transformer.py:55);
  * attention, matching, propagation and the local cost volume are single fused HIP launches
    (``unimatch_amd.ops.HipOps``); no L x L, n x n or [B, L, C, taps] tensor is ever materialised;
  * inference only (the reference's training-mode extra outputs are out of scope).
  * the Transformer's linears / LayerNorm / FFN, the CNN encoder, the refinement block and the upsampler's mask head run
    on the same library (split-fp16 MFMA GEMM / implicit-GEMM convolution kernels, channels-last): on a GPU no MIOpen kernel
    or hipBLASLt kernel is left in the flow, stereo and depth forwards (rocprofv3 kernel statistics of all five configs: profiles/).
There is no CPU or PyTorch fallback for the hot path (the stock ``nn.Module`` convolution code only runs on CPU tensors).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .encoder import CNNEncoder
from .refine import BasicUpdateBlock, convex_upsample
from .refine_nhwc import NhwcUpdateBlock

_IMAGENET_MEAN = (0.485, 0.456, 0.406)
_IMAGENET_STD = (0.229, 0.224, 0.225)


def attention_windows(attn_type, is_self, splits, h, w, with_shift):
    """(win_h, win_w, shift_h, shift_w) of one attention layer.

    Dispatch table of the reference (transformer.py:62-135) with the self/cross role made structural.
    2-D swin: K x K windows; 'row': every scanline; 'winrow': K windows per scanline; 'full': whole map.
    """
    if attn_type == 'swin':
        kind = 'win2d'
    elif attn_type == 'self_swin2d_cross_1d':
        kind = 'win2d' if is_self else 'row'
    elif attn_type == 'self_swin2d_cross_swin1d':
        kind = 'win2d' if is_self else 'winrow'
    else:
        kind = 'full'
    if splits <= 1:
        kind = {'win2d': 'full', 'winrow': 'row'}.get(kind, kind)
    if kind == 'full':
        return h, w, 0, 0
    if kind == 'row':
        return 1, w, 0, 0
    if kind == 'win2d':
        if h % splits or w % splits:
            raise AssertionError(f'feature map {h}x{w} is not divisible by attn_splits={splits}')
        wh, ww = h // splits, w // splits
        return (wh, ww, wh // 2, ww // 2) if with_shift else (wh, ww, 0, 0)
    if w % splits:
        raise AssertionError(f'feature width {w} is not divisible by attn_splits={splits}')
    ww = w // splits
    return (1, ww, 0, ww // 2) if with_shift else (1, ww, 0, 0)


def sine_position_tokens(win_h, win_w, reps_h, reps_w, channels):
    """Token-major ``[reps_h*win_h*reps_w*win_w, C]`` sine position table (fp32, CPU).

    Same arithmetic, in the same order, as the reference's DETR embedding evaluated on a window-sized grid
    (position.py:26-46, utils.py:114-124): normalised to 2*pi with eps 1e-6, temperature 10000, y features in
    the first C/2 channels.  The table is tiled over the reps_h x reps_w windows.
    """
    half = channels // 2
    ys = torch.arange(1, win_h + 1, dtype=torch.float32)
    xs = torch.arange(1, win_w + 1, dtype=torch.float32)
    ys = ys / (ys[-1] + 1e-6) * (2 * math.pi)
    xs = xs / (xs[-1] + 1e-6) * (2 * math.pi)
    j = torch.arange(half, dtype=torch.float32)
    dim_t = 10000.0 ** (2 * torch.div(j, 2, rounding_mode='floor') / half)

    def enc(v):
        ang = v[:, None] / dim_t
        out = torch.empty_like(ang)
        out[:, 0::2] = ang[:, 0::2].sin()
        out[:, 1::2] = ang[:, 1::2].cos()
        return out

    ey, ex = enc(ys), enc(xs)                                   # [win_h, half], [win_w, half]
    tab = torch.cat([ey[:, None, :].expand(win_h, win_w, half), ex[None, :, :].expand(win_h, win_w, half)], -1)
    return tab.repeat(reps_h, reps_w, 1).reshape(-1, channels).contiguous()


class TransformerLayer(nn.Module):
    """Attention (+ optional FFN) layer; parameters as in transformer.py:9-40, forward via HipOps."""

    def __init__(self, d_model, no_ffn, ffn_dim_expansion):
        super().__init__()
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.norm1 = nn.LayerNorm(d_model)
        self.no_ffn = no_ffn
        if not no_ffn:
            self.mlp = nn.Sequential(nn.Linear(2 * d_model, 2 * d_model * ffn_dim_expansion, bias=False), nn.GELU(),
                                     nn.Linear(2 * d_model * ffn_dim_expansion, d_model, bias=False))
            self.norm2 = nn.LayerNorm(d_model)

    def forward(self, ops, source, target, h, w, geom, kv_rotate=0, kv=None, next_kv_weights=None):
        """``kv_rotate = r`` (fused path only): ``target`` holds the streams in the SAME order as ``source`` and stream ``s``
        attends the keys / values of stream ``(s + r) mod S`` -- the swapped copy ``[f1; f0]`` is never built.
        ``kv = (k_operand, v_operand)``: the layer's key / value projections of ``target`` already exist as operand planes
        (``ops.kv4_slices``: the block projects both layers' k | v in one launch, or the previous block's FFN epilogue wrote them)."""
        if getattr(ops, 'fused_tail', False):
            return self._forward_fused(ops, source, target, h, w, geom, kv_rotate, kv, next_kv_weights)
        assert kv_rotate == 0 and kv is None and next_kv_weights is None
        q, k, v = self.q_proj(source), self.k_proj(target), self.v_proj(target)
        msg = ops.window_attention(q, k, v, h, w, *geom)
        msg = self.norm1(self.merge(msg))
        if not self.no_ffn:
            msg = self.norm2(self.mlp(torch.cat([source, msg], dim=-1)))
        return source + msg

    def _forward_fused(self, ops, source, target, h, w, geom, kv_rotate=0, kv=None, next_kv_weights=None):
        """Same layer on the fused HIP path: projections emit attention operand planes, merge + LayerNorm
        (+ residual) is one kernel, the FFN is one kernel (``um_ffn_fwd``; or two without ``ops.fused_ffn``).
        ``next_kv_weights``: the four k | v projection weights of the NEXT block -- the FFN launch then also returns that
        block's blocked k | v planes (``ops.ffn_ln_kv``) and the result is ``(tokens, planes)``."""
        s, l, c = source.shape
        m = s * l
        src = source.reshape(m, c)
        if getattr(ops, 'fused_qproj', False) and getattr(ops, 'fused_merge', False):
            # q is projected inside the attention kernel's prologue (um_window_attn_qproj_merge_fwd): only k | v planes exist
            if kv is None:
                kvp, _, n2 = ops.linear_planes(target.reshape(m, c), (self.k_proj.weight, self.v_proj.weight))
                koff, voff = 0, c
                kop, vop = (kvp, m, n2, koff), (kvp, m, n2, voff)
            else:
                kop, vop = kv                                   # attention operands prepared by the block (ops.kv4_slices)
                assert kop[1] == m
            res = src if self.no_ffn else None
            msg = ops.window_attention_qproj_merge(src, self.q_proj.weight, kop, vop, s, h, w, *geom,
                                                   kv_rotate, self.merge.weight, self.norm1, res)
            if self.no_ffn:
                return msg
            msg = msg.reshape(m, c)
            if next_kv_weights is not None:
                out, nkv = ops.ffn_ln_kv(src, msg, self.mlp[0].weight, self.mlp[2].weight, self.norm2, next_kv_weights)
                return out.reshape(s, l, c), nkv
            if getattr(ops, 'fused_ffn', False):
                return ops.ffn_ln(src, msg, self.mlp[0].weight, self.mlp[2].weight, self.norm2).reshape(s, l, c)
            hid, _, nh = ops.linear_planes(src, (self.mlp[0].weight,), a1=msg, gelu=True)
            return ops.linear_ln(hid, (self.mlp[2].weight,), self.norm2, residual=src, a_planes_k=nh).reshape(s, l, c)
        assert kv is None and next_kv_weights is None, 'precomputed k | v planes need the fused q-projection / merge path'
        if target is source:                       # self attention: one projection launch for q | k | v
            qkv, _, n3 = ops.linear_planes(src, (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight))
            q, k, v = (qkv, m, n3, 0), (qkv, m, n3, c), (qkv, m, n3, 2 * c)
        else:
            qp, _, _ = ops.linear_planes(src, (self.q_proj.weight,))
            kv, _, n2 = ops.linear_planes(target.reshape(m, c), (self.k_proj.weight, self.v_proj.weight))
            q, k, v = (qp, m, c, 0), (kv, m, n2, 0), (kv, m, n2, c)
        if getattr(ops, 'fused_merge', False):      # merge + LayerNorm (+ residual) inside the attention kernel
            res = src if self.no_ffn else None
            msg = ops.window_attention_merge(q, k, v, s, h, w, *geom, kv_rotate, self.merge.weight, self.norm1, res)
            if self.no_ffn:
                return msg
            msg = msg.reshape(m, c)
        else:
            att = ops.window_attention_planes(q, k, v, s, h, w, *geom, kv_rotate=kv_rotate).reshape(m, c)
            if self.no_ffn:
                return ops.linear_ln(att, (self.merge.weight,), self.norm1, residual=src).reshape(s, l, c)
            msg = ops.linear_ln(att, (self.merge.weight,), self.norm1)
        if getattr(ops, 'fused_ffn', False):      # one kernel, the [M, 8C] hidden activations never reach HBM
            return ops.ffn_ln(src, msg, self.mlp[0].weight, self.mlp[2].weight, self.norm2).reshape(s, l, c)
        hid, _, nh = ops.linear_planes(src, (self.mlp[0].weight,), a1=msg, gelu=True)
        return ops.linear_ln(hid, (self.mlp[2].weight,), self.norm2, residual=src, a_planes_k=nh).reshape(s, l, c)


class TransformerBlock(nn.Module):
    def __init__(self, d_model, ffn_dim_expansion):
        super().__init__()
        self.self_attn = TransformerLayer(d_model, True, ffn_dim_expansion)
        self.cross_attn_ffn = TransformerLayer(d_model, False, ffn_dim_expansion)


class FeatureTransformer(nn.Module):
    """6 x (self-attention, cross-attention + FFN) on both images at once (transformer.py:203-294)."""

    def __init__(self, num_layers=6, d_model=128, nhead=1, ffn_dim_expansion=4):
        super().__init__()
        if nhead != 1:
            raise NotImplementedError('multi-head attention is not implemented (neither does the reference, '
                                      'transformer.py:63-66)')
        self.d_model = d_model
        self.layers = nn.ModuleList([TransformerBlock(d_model, ffn_dim_expansion) for _ in range(num_layers)])
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, ops, tok0, tok1, h, w, attn_type, attn_num_splits, stream=None):
        """tok0, tok1: ``[B, h*w, C]`` token-major features (position already added) -- or ``stream`` = the two already
        stacked as ``[2B, h*w, C]`` (what the encoder's batched output is: no concatenation copy)."""
        if stream is None:
            stream = torch.cat([tok0, tok1], 0)        # updated stream  [f0; f1]
        b = stream.shape[0] // 2
        rotate = getattr(ops, 'fused_tail', False)     # cross-attention target [f1; f0] = `prev` read with the halves rotated
        prev = stream if rotate else torch.cat([stream[b:], stream[:b]], 0)
        # both layers of a block read their keys / values from the stream AS IT ENTERS the block (self: the stream itself; cross:
        # its halves rotated), so the four projections Wk_self | Wv_self | Wk_cross | Wv_cross are ONE launch that reads the
        # stream once and writes blocked operand planes [NS][4][M][128] (um_kv4_fwd; round 4 -- was two um_linear_fwd launches of
        # two projections each: transformer.py:58-60 per layer)
        block_kv = rotate and getattr(ops, 'fused_qproj', False) and getattr(ops, 'fused_merge', False) and getattr(ops, 'block_kv', True)
        s2, l, c = stream.shape
        # ... and from block 1 on they come out of the previous block's FFN launch (um_ffn_kv_fwd): no projection launch at all
        fused_kv = block_kv and getattr(ops, 'fused_kv', False) and getattr(ops, 'fused_ffn', False)
        kv4 = None
        kvw = lambda b_: (b_.self_attn.k_proj.weight, b_.self_attn.v_proj.weight, b_.cross_attn_ffn.k_proj.weight,
                          b_.cross_attn_ffn.v_proj.weight)
        for i, blk in enumerate(self.layers):
            shift = ('swin' in attn_type) and attn_num_splits > 1 and i % 2 == 1
            g_self = attention_windows(attn_type, True, attn_num_splits, h, w, shift)
            g_cross = attention_windows(attn_type, False, attn_num_splits, h, w, shift)
            kv_s = kv_c = None
            if block_kv:
                if kv4 is None:                        # block 0 (or fused_kv off): the stand-alone launch
                    kv4 = ops.kv4_planes(stream.reshape(s2 * l, c), kvw(blk))
                kv_s, kv_c = ops.kv4_slices(kv4, s2 * l)
                kv4 = None
            stream = blk.self_attn(ops, stream, stream, h, w, g_self, kv=kv_s)
            if rotate:                                 # keys / values come from the stream as it was before this block
                nxt = kvw(self.layers[i + 1]) if fused_kv and i + 1 < len(self.layers) else None
                stream = blk.cross_attn_ffn(ops, stream, prev, h, w, g_cross, kv_rotate=b, kv=kv_c, next_kv_weights=nxt)
                if nxt is not None:
                    stream, kv4 = stream
                prev = stream
            else:
                stream = blk.cross_attn_ffn(ops, stream, prev, h, w, g_cross)
                prev = torch.cat([stream[b:], stream[:b]], 0)
        return stream[:b].contiguous(), stream[b:].contiguous()


class SelfAttnPropagation(nn.Module):
    """Flow propagation with feature self-similarity (attention.py:166-253); parameters only."""

    def __init__(self, in_channels):
        super().__init__()
        self.q_proj = nn.Linear(in_channels, in_channels)
        self.k_proj = nn.Linear(in_channels, in_channels)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, ops, tok0, flow, h, w, local_window_attn=False, local_window_radius=1):
        if hasattr(ops, 'linear_bias') and tok0.is_cuda:              # the library's own biased Linear (um_linear_bias_fwd)
            if not local_window_attn:
                # reference quirk kept on purpose: the global path projects the *query* again (attention.py:198-205)
                return ops.prop_global_projected(tok0, self.q_proj, self.k_proj, flow, h, w)
            b, l, c = tok0.shape
            x = tok0.reshape(b * l, c)
            q = ops.linear_bias(x, self.q_proj.weight, self.q_proj.bias).view(b, l, c)
            k = ops.linear_bias(x, self.k_proj.weight, self.k_proj.bias).view(b, l, c)
            return ops.prop_local(q, k, flow, h, w, local_window_radius)
        q = self.q_proj(tok0)                                         # injected CPU oracle backend (host-logic tests)
        if local_window_attn:
            return ops.prop_local(q, self.k_proj(tok0), flow, h, w, local_window_radius)
        return ops.prop_global(q, self.k_proj(q), flow, h, w)


def _to_tokens(fmap):
    return fmap.flatten(2).transpose(1, 2).contiguous()


def _to_map(tokens, h, w):
    b, _, c = tokens.shape
    return tokens.transpose(1, 2).reshape(b, c, h, w)


def _pixel_grid(h, w, device):
    ys, xs = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32),
                            torch.arange(w, device=device, dtype=torch.float32), indexing='ij')
    return torch.stack([xs, ys], 0)


def _warp(fmap, flow):
    """Bilinear warp with zero padding, align_corners (geometry.py:41-72); torch op, not on the hot path."""
    b, c, h, w = fmap.shape
    pos = _pixel_grid(h, w, fmap.device)[None] + flow
    grid = torch.stack([2 * pos[:, 0] / (w - 1) - 1, 2 * pos[:, 1] / (h - 1) - 1], -1)
    return F.grid_sample(fmap, grid, mode='bilinear', padding_mode='zeros', align_corners=True)


def _rigid_flow(depth, intrinsics, pose):
    """Flow induced by depth and relative pose (geometry.py:99-195)."""
    b, h, w = depth.shape
    grid = _pixel_grid(h, w, depth.device)
    homog = torch.cat([grid, torch.ones(1, h, w, device=depth.device)], 0).flatten(1)
    pts = (torch.inverse(intrinsics) @ homog) * depth.view(b, 1, -1)
    pts = pose[:, :3, :3] @ pts + pose[:, :3, 3:]
    proj = intrinsics @ pts
    return (proj[:, :2] / proj[:, 2:].clamp(min=1e-3)).view(b, 2, h, w) - grid


class UniMatch(nn.Module):
    def __init__(self, num_scales=1, feature_channels=128, upsample_factor=8, num_head=1, ffn_dim_expansion=4,
                 num_transformer_layers=6, reg_refine=False, task='flow'):
        super().__init__()
        if feature_channels != 128:
            raise NotImplementedError('the HIP kernels are built for feature_channels=128 '
                                      '(the value of every released UniMatch model)')
        self.feature_channels = feature_channels
        self.num_scales = num_scales
        self.upsample_factor = upsample_factor
        self.reg_refine = reg_refine
        self.backbone = CNNEncoder(output_dim=feature_channels, num_output_scales=num_scales)
        self.transformer = FeatureTransformer(num_layers=num_transformer_layers, d_model=feature_channels,
                                              nhead=num_head, ffn_dim_expansion=ffn_dim_expansion)
        self.feature_flow_attn = SelfAttnPropagation(in_channels=feature_channels)
        if not reg_refine or task == 'depth':
            self.upsampler = nn.Sequential(nn.Conv2d(2 + feature_channels, 256, 3, 1, 1), nn.ReLU(inplace=True),
                                           nn.Conv2d(256, upsample_factor ** 2 * 9, 1, 1, 0))
        if reg_refine:
            self.refine_proj = nn.Conv2d(128, 256, 1)
            self.refine = BasicUpdateBlock(corr_channels=(2 * 4 + 1) ** 2, downsample_factor=upsample_factor,
                                           flow_dim=2 if task == 'flow' else 1, bilinear_up=task == 'depth')
        # not part of the state_dict: hot-path backend and caches
        self._ops = None
        self._precision = 'exact'
        self._pos_cache = {}
        self.debug_taps = None            # set to a dict to collect named intermediates (diagnostics only)
        self.check_weights = False        # True: fingerprint the parameters every forward (see _check_weight_print)
        self._weight_print = None
        self._runner = None               # streams.PartRunner: the batch as concurrent forwards (see forward)

    # ------------------------------------------------------------------ hot-path backend
    def set_precision(self, precision):
        """'exact' (fp16 hi+lo split MFMA operands; parity mode) or 'fast' (bf16 operands)."""
        self._precision = precision
        self._ops = None
        return self

    def invalidate_weights(self):
        """Drop the cached MFMA operand planes of the weights.  Needed only after an in-place edit through ``.data``
        (``p.data.mul_(2)`` changes neither the tensor's identity, version nor address); ``load_state_dict``, ``.to()`` /
        ``.float()`` / ``.cuda()`` and ``p.data = new`` are detected without it."""
        if self._ops is not None:
            self._ops.invalidate_weights()
        self._weight_print = None
        return self

    def _apply(self, fn, *args, **kwargs):            # .to() / .cuda() / .float(): parameters move, caches must not survive
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_weights()
        self._pos_cache = {}
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_weights()
        return out

    def _check_weight_print(self):
        """check_weights=True: a per-forward fingerprint of all parameters (one fused norm, one host sync) that catches
        in-place ``.data`` edits automatically -- off by default because the sync serialises back-to-back forwards."""
        params = [p.detach() for p in self.parameters()]
        fp = torch.stack(torch._foreach_norm(params)).double().cpu()
        if self._weight_print is not None and not torch.equal(fp, self._weight_print):
            if self._ops is not None:
                self._ops.invalidate_weights()
        self._weight_print = fp

    def bind_ops(self, ops):
        """Install a hot-path backend explicitly (tests inject the CPU oracle here)."""
        self._ops = ops
        return self

    @property
    def ops(self):
        if self._ops is None:
            from .ops import HipOps           # raises when the HIP extension or the GPU is missing
            self._ops = HipOps(self._precision)
        return self._ops

    def _position(self, h, w, splits, device):
        # same precondition as the reference's split_feature (unimatch/utils.py:40)
        assert splits <= 1 or (h % splits == 0 and w % splits == 0), \
            f'feature map {h}x{w} is not divisible by attn_splits={splits}'
        key = (h, w, splits, str(device))
        if key not in self._pos_cache:
            if splits > 1:
                tab = sine_position_tokens(h // splits, w // splits, splits, splits, self.feature_channels)
            else:
                tab = sine_position_tokens(h, w, 1, 1, self.feature_channels)
            self._pos_cache[key] = tab.to(device)
        return self._pos_cache[key]

    def _constants(self, device):
        """ImageNet mean / std on ``device``, uploaded once (keeps the forward free of host->device copies so that
        it can be captured into a HIP graph)."""
        key = ('imagenet', str(device))
        if key not in self._pos_cache:
            self._pos_cache[key] = (torch.tensor(_IMAGENET_MEAN).view(1, 3, 1, 1).to(device),
                                    torch.tensor(_IMAGENET_STD).view(1, 3, 1, 1).to(device))
        return self._pos_cache[key]

    def _depth_candidates(self, lo, hi, n, device):
        """Inverse-depth candidates: computed on the CPU exactly as the reference does (unimatch.py:187), cached."""
        key = ('cand', float(lo), float(hi), int(n), str(device))
        if key not in self._pos_cache:
            self._pos_cache[key] = torch.linspace(lo, hi, n).float().to(device).contiguous()
        return self._pos_cache[key]

    def _convex(self, flow2, mask, is_depth=False):
        ops = self.ops
        if hasattr(ops, 'convex_upsample') and flow2.is_cuda and self.upsample_factor in (4, 8):
            return ops.convex_upsample(flow2, mask, self.upsample_factor, is_depth)
        return convex_upsample(flow2, mask, self.upsample_factor, is_depth=is_depth)

    check_range = True       # read the operand-range flags at the start of every exact-mode forward (no synchronisation)

    def check_operand_range(self, sync=True):
        """Raise ``_abi.OperandRangeError`` if any exact-mode kernel met an activation outside fp16's range since the last check
        (``sync``: wait for the device first, so that the forward that has just been enqueued is covered)."""
        from . import _abi
        if sync and torch.cuda.is_available():
            torch.cuda.synchronize()
        _abi.check_operand_range()

    def _upsample_mask(self, flow2, f0_map):
        """The upsampler head (unimatch.py:56-58): convex-combination logits of the full-resolution prediction ->
        ``(mask, mask_is_nhwc)``; on the GPU the two convolutions run channels-last on the library's kernels."""
        ops = self.ops
        if getattr(ops, 'fused_conv', False) and flow2.is_cuda and self.upsample_factor in (4, 8):
            # cat(flow, feature) -> 3x3 + ReLU -> 1x1 -> NHWC mask
            b, v, h, w = flow2.shape
            rows = b * h * w
            feat = f0_map.permute(0, 2, 3, 1).reshape(rows, -1)          # a view when f0_map came from tokens
            planes, cin = ops.nhwc_planes_from([flow2.permute(0, 2, 3, 1).reshape(rows, v), feat])
            c1, c2 = self.upsampler[0], self.upsampler[2]
            hid, _, _ = ops.conv2d_nhwc((planes, b, h, w, cin), ops.conv_weight_padded(c1.weight, cin), c1.bias, 1, (1, 1),
                                        relu=True)
            hp, hc = ops.nhwc_planes_from([hid])
            mask, _, _ = ops.conv2d_nhwc((hp, b, h, w, hc), c2.weight, c2.bias, 1, (0, 0))
            return mask, True
        return self.upsampler(torch.cat([flow2, f0_map], 1)), False

    def _upsample(self, flow2, f0_map, is_depth=False):
        mask, nhwc = self._upsample_mask(flow2, f0_map)
        if nhwc:
            return self.ops.convex_upsample(flow2, mask, self.upsample_factor, is_depth, mask_nhwc=True)
        return self._convex(flow2, mask, is_depth=is_depth)

    # ------------------------------------------------------------------ forward
    launch_parts = None      # None: streams.forward_parts() decides per call; an int forces that many concurrent forwards (1: never split)

    def forward(self, img0, img1, attn_type=None, attn_splits_list=None, corr_radius_list=None,
                prop_radius_list=None, num_reg_refine=1, pred_bidir_flow=False, task='flow', intrinsics=None,
                pose=None, min_depth=1. / 0.5, max_depth=1. / 10, num_depth_candidates=64,
                depth_from_argmax=False, pred_bidir_depth=False, **kwargs):
        """The reference's entry point (unimatch/unimatch.py:95-111), the only one a caller such as evaluate_flow.py:405-412 knows.

        Round 6: HOW the batch is launched is decided here, by a pure function of the call (``streams.forward_parts``): either one
        forward of the whole batch, or -- where the launches of one forward leave tails that a second forward fills (measured per
        configuration) -- the batch as two forwards of contiguous sample ranges on two HIP streams, joined on the caller's
        stream.  The samples of a batch are independent (unimatch.py:113-367 has no cross-sample operation), every part is bitwise
        the plain forward of its samples."""
        if self.training:
            raise RuntimeError('this module implements inference only: call .eval() '
                               '(training-mode auxiliary outputs of the reference are out of scope)')
        kw = dict(attn_type=attn_type, attn_splits_list=attn_splits_list, corr_radius_list=corr_radius_list,
                  prop_radius_list=prop_radius_list, num_reg_refine=num_reg_refine, pred_bidir_flow=pred_bidir_flow, task=task,
                  intrinsics=intrinsics, pose=pose, min_depth=min_depth, max_depth=max_depth,
                  num_depth_candidates=num_depth_candidates, depth_from_argmax=depth_from_argmax, pred_bidir_depth=pred_bidir_depth)
        parts = self.launch_parts
        if parts is None:
            parts = 1
            if img0.is_cuda and self.debug_taps is None:
                from .streams import forward_parts
                parts = forward_parts(task, attn_type, self.num_scales, self.reg_refine, img0.shape[0], img0.shape[-2], img0.shape[-1])
        parts = min(int(parts), img0.shape[0])
        if parts <= 1:
            return self._forward_one(img0, img1, **kw)
        if self._runner is None:
            from .streams import PartRunner
            self._runner = PartRunner()
        return self._runner.run(self, parts, img0, img1, kw)

    def _forward_one(self, img0, img1, attn_type=None, attn_splits_list=None, corr_radius_list=None,
                     prop_radius_list=None, num_reg_refine=1, pred_bidir_flow=False, task='flow', intrinsics=None,
                     pose=None, min_depth=1. / 0.5, max_depth=1. / 10, num_depth_candidates=64,
                     depth_from_argmax=False, pred_bidir_depth=False):
        """One forward of the given samples on the current stream."""
        if pred_bidir_flow:
            assert task == 'flow'
        if task == 'depth':
            assert self.num_scales == 1
            assert len(attn_splits_list) == len(prop_radius_list) == self.num_scales == 1
        else:
            assert len(attn_splits_list) == len(corr_radius_list) == len(prop_radius_list) == self.num_scales
        ops = self.ops
        if self.check_weights:
            self._check_weight_print()
        attn_type = attn_type if attn_type is not None else ''
        dev = img0.device
        # Operand range of exact mode (include/unimatch_hip.h): the kernels raise a sticky flag in pinned host memory when an
        # activation >= 65504 is turned into an fp16 operand.  Reading it costs no synchronisation, so it is read HERE, at the start
        # of every forward: an overflow of an earlier (finished) forward is reported loudly instead of living on as NaN predictions;
        # `check_operand_range()` synchronises first and covers the call that has just been made.
        if img0.is_cuda and self.check_range and getattr(ops, 'mode', 1) == 0:
            from . import _abi
            _abi.check_operand_range('an earlier forward of this process: ')

        # every launch of the library goes to torch's CURRENT stream and scratch buffers are allocated on the current device:
        # make the inputs' device current for the whole forward (a model on cuda:1 called while cuda:0 is current)
        import contextlib
        guard = torch.cuda.device(dev) if img0.is_cuda else contextlib.nullcontext()
        if img1.device != dev:
            raise ValueError(f'img0 is on {dev} and img1 on {img1.device}')
        with guard, torch.no_grad():
            input_norm = None
            if task == 'flow':                  # stereo / depth loaders normalise already (unimatch.py:122-124)
                if self.backbone.takes_raw_images(ops, img0):
                    input_norm = (_IMAGENET_MEAN, _IMAGENET_STD)            # folded into the stem's image packing
                else:
                    mean, std = self._constants(dev)
                    img0, img1 = (img0 / 255. - mean) / std, (img1 / 255. - mean) / std
            feats = self.backbone(torch.cat([img0, img1], 0), ops, input_norm)[::-1]    # low -> high resolution
            nb = img0.shape[0]
            flow, pred = None, None
            for s in range(self.num_scales):
                m0, m1 = feats[s][:nb], feats[s][nb:]                         # [B, C, h, w]
                both = None
                if pred_bidir_flow and s > 0:
                    m0, m1 = torch.cat([m0, m1], 0), torch.cat([m1, m0], 0)
                    ori0, ori1 = _to_tokens(m0), _to_tokens(m1)               # pre-position, pre-warp tokens
                else:
                    both = _to_tokens(feats[s])                               # the encoder's batched output is [f0; f1]
                    ori0, ori1 = both[:nb], both[nb:]
                h, w = m0.shape[-2:]
                up = self.upsample_factor * 2 ** (self.num_scales - 1 - s)
                if task == 'depth':
                    k_cur = intrinsics.clone()
                    k_cur[:, :2] = k_cur[:, :2] / up
                if s > 0:
                    if hasattr(ops, 'flow_upsample2x') and flow.is_cuda:
                        flow = ops.flow_upsample2x(flow, 2.0)          # unimatch.py:162-163 on um_flow_upsample2x
                    else:
                        flow = F.interpolate(flow, scale_factor=2, mode='bilinear', align_corners=True) * 2
                tok1 = ori1
                if flow is not None:
                    assert task != 'depth'
                    disp = torch.cat([-flow, torch.zeros_like(flow)], 1) if task == 'stereo' else flow
                    if hasattr(ops, 'flow_warp') and ori1.is_cuda:
                        tok1 = ops.flow_warp(ori1, disp.contiguous(), h, w)
                    else:
                        tok1 = _to_tokens(_warp(m1, disp))
                splits, prop_r = attn_splits_list[s], prop_radius_list[s]
                pos = self._position(h, w, splits, dev)
                if self.debug_taps is not None:
                    self.debug_taps[f'backbone0_s{s}'], self.debug_taps[f'backbone1_s{s}'] = m0, m1
                if tok1 is ori1 and both is not None:
                    # no warp, no bidirectional stacking: the stream [f0; f1] is the encoder's output -- one position add on
                    # the whole of it instead of two adds and a concatenation
                    tok0, tok1 = self.transformer(ops, None, None, h, w, attn_type, splits, stream=both + pos)
                else:
                    tok0, tok1 = self.transformer(ops, ori0 + pos, tok1 + pos, h, w, attn_type, splits)
                if self.debug_taps is not None:
                    self.debug_taps[f'f0_s{s}'], self.debug_taps[f'f1_s{s}'] = _to_map(tok0, h, w), _to_map(tok1, h, w)

                # ---- matching layer
                if task == 'depth':
                    cand = self._depth_candidates(min_depth, max_depth, num_depth_candidates, dev)
                    f0c, f1c = tok0, tok1
                    if pred_bidir_depth:
                        f0c, f1c = torch.cat([tok0, tok1], 0), torch.cat([tok1, tok0], 0)
                    if hasattr(ops, 'depth_cam') and tok0.is_cuda:     # K / up, K^-1, (inverse) pose packed on the device, no sync
                        cam = ops.depth_cam(intrinsics, pose, float(up), pred_bidir_depth)
                    else:
                        kc, pc = k_cur, pose
                        if pred_bidir_depth:
                            kc = k_cur.repeat(2, 1, 1)
                            pc = torch.cat([pose, torch.inverse(pose)], 0)
                        cam = torch.cat([torch.inverse(kc).flatten(1), pc[:, :3, :3].flatten(1), pc[:, :3, 3],
                                         kc.flatten(1)], 1).float().contiguous()
                    flow_pred = ops.depth_corr_softmax(f0c, f1c, h, w, cam, cand.contiguous(), depth_from_argmax)
                else:
                    radius = corr_radius_list[s]
                    if radius == -1:
                        if task == 'flow':
                            flow_pred = ops.global_corr_softmax_flow(tok0, tok1, h, w, pred_bidir_flow)
                        elif task == 'stereo':
                            flow_pred = ops.global_corr_softmax_stereo(tok0, tok1, h, w)
                        else:
                            raise NotImplementedError
                    else:
                        if task not in ('flow', 'stereo'):
                            raise NotImplementedError
                        flow_pred = ops.local_corr_softmax(tok0, tok1, h, w, radius, one_d=(task == 'stereo'))
                flow = flow_pred if flow is None else flow + flow_pred
                if task == 'stereo':
                    flow = flow.clamp(min=0)
                if self.debug_taps is not None:
                    self.debug_taps[f'flow_match_s{s}'] = flow

                # ---- propagation
                if (pred_bidir_flow or pred_bidir_depth) and s == 0:
                    tok0 = torch.cat([tok0, tok1], 0)
                flow = self.feature_flow_attn(ops, tok0, flow.contiguous(), h, w,
                                              local_window_attn=prop_r > 0, local_window_radius=prop_r)
                if self.debug_taps is not None:
                    self.debug_taps[f'flow_prop_s{s}'] = flow
                if s < self.num_scales - 1:
                    continue

                # ---- full-resolution prediction
                f0_map = _to_map(tok0, h, w)
                if not self.reg_refine:
                    if task == 'stereo':
                        pad = torch.cat([-flow, torch.zeros_like(flow)], 1)
                        pred = -self._upsample(pad, f0_map)[:, :1]
                    elif task == 'depth':
                        pad = torch.cat([flow, torch.zeros_like(flow)], 1)
                        pred = self._upsample(pad, f0_map, is_depth=True).clamp(min=min_depth, max=max_depth)[:, :1]
                    else:
                        pred = self._upsample(flow, f0_map)
                    continue
                assert num_reg_refine > 0
                pose_r = pose
                nhwc = None                                     # channels-last refinement block on the library's convolutions
                if getattr(ops, 'fused_conv', False) and tok0.is_cuda:      # every task: flow_dim 2 (flow) / 1 (disparity, inverse depth)
                    nhwc = NhwcUpdateBlock(ops, self.refine, self.refine_proj)
                    nhwc.begin(tok0, tok0.shape[0], h, w, iterations=num_reg_refine)
                else:
                    proj = self.refine_proj(f0_map)             # same every iteration (unimatch.py:315-320)
                    net0, inp = torch.tanh(proj[:, :128]), torch.relu(proj[:, 128:])
                for it in range(num_reg_refine):
                    if task == 'stereo':
                        disp = torch.cat([-flow, torch.zeros_like(flow)], 1)
                    elif task == 'depth':
                        if pred_bidir_depth and it == 0:
                            ori0, ori1 = torch.cat([ori0, ori1], 0), torch.cat([ori1, ori0], 0)
                        if hasattr(ops, 'rigid_flow') and flow.is_cuda:     # geometry.py:99-195 on um_rigid_flow (cam from the matching layer)
                            disp = ops.rigid_flow(flow, cam)
                        else:
                            if pred_bidir_depth and it == 0:
                                k_cur = k_cur.repeat(2, 1, 1)
                                pose_r = torch.cat([pose, torch.inverse(pose)], 0)
                            disp = _rigid_flow(1. / flow.squeeze(1), k_cur, pose_r)
                    else:
                        disp = flow
                    if nhwc is not None:
                        up_mask, delta = nhwc.iterate(ori0, ori1, disp.contiguous(), flow, it == num_reg_refine - 1)
                    else:
                        corr = ops.local_corr_with_flow(ori0, ori1, disp.contiguous(), h, w, 4)
                        _, up_mask, delta = self.refine(net0, inp, corr, flow)
                    if task == 'depth':
                        flow = (flow - delta).clamp(min=min_depth, max=max_depth)
                    else:
                        flow = flow + delta
                    if task == 'stereo':
                        flow = flow.clamp(min=0)
                    if self.debug_taps is not None:
                        self.debug_taps[f'flow_it{it}'] = flow
                    if it == num_reg_refine - 1:
                        if task == 'depth':
                            pad = torch.cat([flow, torch.zeros_like(flow)], 1)
                            pred = self._upsample(pad, f0_map, is_depth=True).clamp(
                                min=min_depth, max=max_depth)[:, :1]
                        elif nhwc is not None:
                            pred = ops.convex_upsample(flow, up_mask, self.upsample_factor, False, mask_nhwc=True)
                        else:
                            pred = self._convex(flow, up_mask)
            if task == 'stereo':
                pred = pred.squeeze(1)
            if task == 'depth':
                pred = 1. / pred.squeeze(1)
        return {'flow_preds': [pred]}
