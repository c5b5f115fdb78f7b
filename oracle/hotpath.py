"""CPU oracle for the UniMatch global-matching hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, from the maths, what the reference computes on the hot path.  It is *not*
the product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  The product (``unimatch_amd``) never routes through it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference from
``/root/reference`` in the build container and writes small seeded fixtures (``tests/golden/*.npz``);
``tests/test_oracle_vs_golden.py`` checks every function below against them.

All functions are dtype generic (run them in float64 to get a ground truth that is *more* accurate
than the fp32 reference) and operate on torch CPU tensors.  Layout conventions follow the reference:
features are ``[B, C, h, w]``, tokens are row-major ``(y, x)``, flow channel 0 is x.

Everything here is written as *index arithmetic* on token coordinates (no roll / split / mask
tensors), which is also how the HIP kernels address memory — so the oracle doubles as an executable
specification of the kernels' addressing.

Reference citations (relative to /root/reference):
  position table            unimatch/position.py:26-46, unimatch/utils.py:111-131
  window attention (2-D)    unimatch/attention.py:45-104, unimatch/utils.py:34-59,84-108
  window attention (1-D)    unimatch/attention.py:19-42,107-163, unimatch/utils.py:199-216
  transformer layer / stack unimatch/transformer.py:42-144, 226-294
  global flow matching      unimatch/matching.py:7-36
  local flow matching       unimatch/matching.py:39-83
  local cost volume         unimatch/matching.py:86-123
  global / local stereo     unimatch/matching.py:126-151, 154-200
  plane-sweep depth         unimatch/matching.py:203-282
  propagation               unimatch/attention.py:184-253
"""
import math

import torch
import torch.nn.functional as F

MASK_NEG = -100.0        # shifted-window mask value (unimatch/utils.py:106): added, not -inf
OOB_NEG = -1e9           # out-of-image / causal mask value (unimatch/matching.py:73,144)


# ----------------------------------------------------------------------------------------------
# position embedding (unimatch/position.py:26-46) on a window-sized grid (unimatch/utils.py:114-124)
# ----------------------------------------------------------------------------------------------
def position_table(win_h, win_w, channels, dtype=torch.float32):
    """Sine table ``[C, win_h, win_w]``: first C/2 channels encode y, last C/2 encode x."""
    half = channels // 2
    two_pi = 2 * math.pi
    # the reference does this arithmetic in fp32 (cumsum of ones, eps=1e-6); keep the same op order
    # in fp32 so that the table is bit-identical, then cast.
    ys = torch.arange(1, win_h + 1, dtype=torch.float32)
    xs = torch.arange(1, win_w + 1, dtype=torch.float32)
    ys = ys / (ys[-1] + 1e-6) * two_pi
    xs = xs / (xs[-1] + 1e-6) * two_pi
    j = torch.arange(half, dtype=torch.float32)
    dim_t = 10000.0 ** (2 * torch.div(j, 2, rounding_mode='floor') / half)
    py = ys[:, None] / dim_t          # [win_h, half]
    px = xs[:, None] / dim_t          # [win_w, half]

    def interleave(p):                # even slots sin, odd slots cos
        out = torch.empty_like(p)
        out[:, 0::2] = p[:, 0::2].sin()
        out[:, 1::2] = p[:, 1::2].cos()
        return out

    py, px = interleave(py), interleave(px)
    tab = torch.empty(channels, win_h, win_w, dtype=torch.float32)
    tab[:half] = py.t()[:, :, None].expand(half, win_h, win_w)
    tab[half:] = px.t()[:, None, :].expand(half, win_h, win_w)
    return tab.to(dtype)


def add_position(feature0, feature1, splits):
    """Add the per-window sine table to both feature maps (unimatch/utils.py:111-131)."""
    b, c, h, w = feature0.shape
    if splits > 1:
        win_h, win_w = h // splits, w // splits
        tab = position_table(win_h, win_w, c, feature0.dtype).repeat(1, splits, splits)
    else:
        tab = position_table(h, w, c, feature0.dtype)
    return feature0 + tab, feature1 + tab


# ----------------------------------------------------------------------------------------------
# windowed attention by index arithmetic
# ----------------------------------------------------------------------------------------------
def window_index(h, w, win_h, win_w, shift_h, shift_w):
    """Token index and mask label per (window, local token).

    A window of the *rolled* image collects rolled coordinates (ry, rx); the token that sits there
    is the original token ((ry+shift_h) % h, (rx+shift_w) % w) (torch.roll by -shift,
    unimatch/attention.py:72-79).  Labels follow unimatch/utils.py:90-100: 3 row bands x 3 col bands
    on rolled coordinates.  Returns ``idx [n_win, n] (long)``, ``label [n_win, n] (long)``;
    window order is (wy, wx), local order (ly, lx).
    """
    ny, nx = h // win_h, w // win_w
    ry = torch.arange(h).view(ny, 1, win_h, 1).expand(ny, nx, win_h, win_w)
    rx = torch.arange(w).view(1, nx, 1, win_w).expand(ny, nx, win_h, win_w)
    oy = (ry + shift_h) % h
    ox = (rx + shift_w) % w
    idx = (oy * w + ox).reshape(ny * nx, win_h * win_w)
    row_lab = (ry >= h - win_h).long() + (ry >= h - shift_h).long() if shift_h > 0 else torch.zeros_like(ry)
    col_lab = (rx >= w - win_w).long() + (rx >= w - shift_w).long() if shift_w > 0 else torch.zeros_like(rx)
    label = (3 * row_lab + col_lab).reshape(ny * nx, win_h * win_w)
    return idx, label


def window_attention(q, k, v, h, w, win_h, win_w, shift_h=0, shift_w=0):
    """``softmax(q k^T / sqrt(C) + mask) v`` inside windows.  q, k, v: ``[S, L, C]`` tokens.

    Covers every attention flavour of the reference with one parametrisation:
      * 2-D swin (attention.py:45-104):   win = (h/K, w/K), shift = win//2 or 0
      * full 2-D (attention.py:8-16):     win = (h, w), shift 0
      * 1-D per row (attention.py:19-42): win = (1, w), shift 0
      * 1-D swin (attention.py:107-163):  win = (1, w/K), shift = (0, win_w//2) or 0
    """
    s, l, c = q.shape
    idx, label = window_index(h, w, win_h, win_w, shift_h, shift_w)
    nwin, n = idx.shape
    flat = idx.reshape(-1)
    qw = q[:, flat].reshape(s * nwin, n, c)
    kw = k[:, flat].reshape(s * nwin, n, c)
    vw = v[:, flat].reshape(s * nwin, n, c)
    scores = torch.bmm(qw, kw.transpose(1, 2)) / (c ** 0.5)
    if shift_h > 0 or shift_w > 0:
        neq = label[:, :, None] != label[:, None, :]
        mask = torch.zeros(nwin, n, n, dtype=q.dtype).masked_fill_(neq, MASK_NEG)
        scores = (scores.view(s, nwin, n, n) + mask).view(s * nwin, n, n)
    out_w = torch.bmm(torch.softmax(scores, dim=-1), vw).reshape(s, nwin * n, c)
    out = torch.empty_like(q)
    out[:, flat] = out_w
    return out


def attention_geometry(attn_type, is_self, splits, h, w, with_shift):
    """(win_h, win_w, shift_h, shift_w) for a layer, by the *structural* role of the layer.

    The reference decides self-vs-cross from the data (transformer.py:55); structurally the first
    layer of a block is self attention and the second is cross attention (transformer.py:180-200).
    Dispatch table: transformer.py:62-135.
    """
    if attn_type == 'swin':
        kind = 'win2d'
    elif attn_type == 'self_swin2d_cross_1d':
        kind = 'win2d' if is_self else 'row'
    elif attn_type == 'self_swin2d_cross_swin1d':
        kind = 'win2d' if is_self else 'winrow'
    else:
        kind = 'full'
    if splits <= 1:                       # no windows: 2-D kinds degrade to full, 1-D kinds to rows
        kind = {'win2d': 'full', 'winrow': 'row'}.get(kind, kind)
    if kind == 'full':
        return h, w, 0, 0
    if kind == 'row':
        return 1, w, 0, 0
    if kind == 'win2d':
        win_h, win_w = h // splits, w // splits
        return (win_h, win_w, win_h // 2, win_w // 2) if with_shift else (win_h, win_w, 0, 0)
    win_w = w // splits                   # 'winrow'
    return (1, win_w, 0, win_w // 2) if with_shift else (1, win_w, 0, 0)


def transformer_layer(source, target, p, prefix, h, w, geom, with_ffn):
    """One TransformerLayer (transformer.py:42-144).  p: state dict, prefix e.g. 'layers.0.self_attn.'"""
    q = source @ p[prefix + 'q_proj.weight'].t()
    k = target @ p[prefix + 'k_proj.weight'].t()
    v = target @ p[prefix + 'v_proj.weight'].t()
    msg = window_attention(q, k, v, h, w, *geom)
    msg = msg @ p[prefix + 'merge.weight'].t()
    c = msg.shape[-1]
    msg = F.layer_norm(msg, (c,), p[prefix + 'norm1.weight'], p[prefix + 'norm1.bias'])
    if with_ffn:
        x = torch.cat([source, msg], dim=-1) @ p[prefix + 'mlp.0.weight'].t()
        x = F.gelu(x) @ p[prefix + 'mlp.2.weight'].t()
        msg = F.layer_norm(x, (c,), p[prefix + 'norm2.weight'], p[prefix + 'norm2.bias'])
    return source + msg


def transformer_block(a, bt, p, i, attn_type, splits, h, w, prefix=''):
    """Block ``i`` of the stack on the stream ``a = [f0; f1]`` (tokens) with cross-attention target ``bt = [f1; f0]`` as it was
    BEFORE the block (transformer.py:271-291) -> the updated stream."""
    with_shift = ('swin' in attn_type) and splits > 1 and i % 2 == 1
    g_self = attention_geometry(attn_type, True, splits, h, w, with_shift)
    g_cross = attention_geometry(attn_type, False, splits, h, w, with_shift)
    lp = f'{prefix}layers.{i}.'
    a = transformer_layer(a, a, p, lp + 'self_attn.', h, w, g_self, with_ffn=False)
    return transformer_layer(a, bt, p, lp + 'cross_attn_ffn.', h, w, g_cross, with_ffn=True)


def feature_transformer(feature0, feature1, p, attn_type, splits, num_layers=6, prefix='', taps=None, tag=''):
    """FeatureTransformer.forward (transformer.py:226-294).  Returns updated (feature0, feature1).
    ``taps`` (dict): receives the token stream ``[f0; f1]`` entering the stack (``xin{tag}``) and leaving block i (``blk{i}{tag}``)."""
    b, c, h, w = feature0.shape
    t0 = feature0.flatten(2).transpose(1, 2)
    t1 = feature1.flatten(2).transpose(1, 2)
    a = torch.cat([t0, t1], 0)            # stream being updated
    bt = torch.cat([t1, t0], 0)           # its cross-attention target
    if taps is not None:
        taps[f'xin{tag}'] = a
    for i in range(num_layers):
        with_shift = ('swin' in attn_type) and splits > 1 and i % 2 == 1
        g_self = attention_geometry(attn_type, True, splits, h, w, with_shift)
        g_cross = attention_geometry(attn_type, False, splits, h, w, with_shift)
        lp = f'{prefix}layers.{i}.'
        a = transformer_layer(a, a, p, lp + 'self_attn.', h, w, g_self, with_ffn=False)
        a = transformer_layer(a, bt, p, lp + 'cross_attn_ffn.', h, w, g_cross, with_ffn=True)
        bt = torch.cat([a[b:], a[:b]], 0)
        if taps is not None:
            taps[f'blk{i}{tag}'] = a
    f0 = a[:b].transpose(1, 2).reshape(b, c, h, w)
    f1 = a[b:].transpose(1, 2).reshape(b, c, h, w)
    return f0, f1


# ----------------------------------------------------------------------------------------------
# matching layers
# ----------------------------------------------------------------------------------------------
def pixel_grid(h, w, dtype):
    """``[2, h, w]``: channel 0 = x, channel 1 = y (geometry.py:5-21)."""
    ys, xs = torch.meshgrid(torch.arange(h, dtype=dtype), torch.arange(w, dtype=dtype), indexing='ij')
    return torch.stack([xs, ys], 0)


def global_corr_softmax_flow(feature0, feature1, bidir=False):
    """matching.py:7-36: expected matching coordinate under softmax over all target pixels - own."""
    b, c, h, w = feature0.shape
    t0 = feature0.flatten(2).transpose(1, 2)
    t1 = feature1.flatten(2).transpose(1, 2)
    corr = torch.bmm(t0, t1.transpose(1, 2)) / (c ** 0.5)
    if bidir:
        corr = torch.cat([corr, corr.transpose(1, 2)], 0)
    grid = pixel_grid(h, w, feature0.dtype).flatten(1).t()          # [L, 2]
    prob = torch.softmax(corr, dim=-1)
    match = prob @ grid                                             # [B', L, 2]
    flow = (match - grid).transpose(1, 2).reshape(-1, 2, h, w)
    return flow


def _shifted(feature, dy, dx):
    """feature sampled at (y+dy, x+dx) with zeros outside, plus the in-image validity mask."""
    b, c, h, w = feature.shape
    out = torch.zeros_like(feature)
    y0, y1 = max(0, -dy), min(h, h - dy)
    x0, x1 = max(0, -dx), min(w, w - dx)
    valid = torch.zeros(h, w, dtype=torch.bool)
    if y1 > y0 and x1 > x0:
        out[:, :, y0:y1, x0:x1] = feature[:, :, y0 + dy:y1 + dy, x0 + dx:x1 + dx]
        valid[y0:y1, x0:x1] = True
    return out, valid


def local_corr_softmax(feature0, feature1, radius, one_d=False):
    """matching.py:39-83 (2-D) and :154-200 (1-D stereo, returns -flow_x as ``[B,1,h,w]``).

    Taps d=(dx,dy), dy outer, both in [-r, r]; out-of-image taps get logit -1e9.
    """
    b, c, h, w = feature0.shape
    dys = [0] if one_d else list(range(-radius, radius + 1))
    dxs = list(range(-radius, radius + 1))
    logits, offs = [], []
    for dy in dys:
        for dx in dxs:
            f1s, valid = _shifted(feature1, dy, dx)
            lg = (feature0 * f1s).sum(1) / (c ** 0.5)
            lg = torch.where(valid, lg, torch.full_like(lg, OOB_NEG))
            logits.append(lg)
            offs.append((dx, dy))
    logits = torch.stack(logits, 1)                                  # [B, T, h, w]
    prob = torch.softmax(logits, dim=1)
    off = torch.tensor(offs, dtype=feature0.dtype)                   # [T, 2]
    flow = torch.einsum('bthw,tc->bchw', prob, off)
    if one_d:
        return -flow[:, :1]
    return flow


def local_corr_with_flow_dilated(feature0, feature1, flow, radius, dilation):
    """matching.py:86-123 with its ``dilation`` argument, in the reference's own form: one bilinear sample (zeros padding,
    align_corners) per tap at ``p + dilation * d_k + flow(p)``; taps dy-outer / dx-inner."""
    b, c, h, w = feature0.shape
    grid = pixel_grid(h, w, feature0.dtype)
    out = []
    for dy in range(-radius, radius + 1):
        for dx in range(-radius, radius + 1):
            pos = grid[None] + flow + torch.tensor([dx * dilation, dy * dilation], dtype=feature0.dtype).view(1, 2, 1, 1)
            norm = torch.stack([2 * pos[:, 0] / (w - 1) - 1, 2 * pos[:, 1] / (h - 1) - 1], -1)
            samp = torch.nn.functional.grid_sample(feature1, norm, mode='bilinear', padding_mode='zeros', align_corners=True)
            out.append((feature0 * samp).sum(1) / c ** 0.5)
    return torch.stack(out, 1)


def local_corr_with_flow(feature0, feature1, flow, radius):
    """matching.py:86-123: ``out[k,p] = f0(p) . bilinear(f1, p + d_k + flow(p)) / sqrt(C)``, zeros outside.

    All (2r+1)^2 taps of a pixel share one fractional offset, and bilinear sampling is linear in
    feature1, so this is evaluated from the (2r+2)^2 integer-position dot products around
    floor(p+flow) blended with the 4 bilinear weights — same maths, different association.
    """
    b, c, h, w = feature0.shape
    k = 2 * radius + 1
    grid = pixel_grid(h, w, feature0.dtype)
    pos = grid[None] + flow                                           # [B, 2, h, w]
    base = torch.floor(pos)
    frac = pos - base
    bx = base[:, 0].long()
    by = base[:, 1].long()
    f1 = feature1.permute(0, 2, 3, 1)                                 # [B, h, w, C]
    f0 = feature0.permute(0, 2, 3, 1)
    bi = torch.arange(b).view(b, 1, 1)
    n = k + 1
    dots = torch.zeros(b, n, n, h, w, dtype=feature0.dtype)           # integer-neighbourhood dots
    for iy in range(n):
        for ix in range(n):
            yy = by + (iy - radius)
            xx = bx + (ix - radius)
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            g = f1[bi, yy.clamp(0, h - 1), xx.clamp(0, w - 1)]        # [B, h, w, C]
            d = (f0 * g).sum(-1)
            dots[:, iy, ix] = torch.where(ok, d, torch.zeros_like(d))
    fx, fy = frac[:, 0], frac[:, 1]
    w00 = (1 - fx) * (1 - fy)
    w01 = fx * (1 - fy)
    w10 = (1 - fx) * fy
    w11 = fx * fy
    out = torch.empty(b, k * k, h, w, dtype=feature0.dtype)
    for ty in range(k):
        for tx in range(k):
            val = (w00 * dots[:, ty, tx] + w01 * dots[:, ty, tx + 1]
                   + w10 * dots[:, ty + 1, tx] + w11 * dots[:, ty + 1, tx + 1])
            out[:, ty * k + tx] = val / (c ** 0.5)
    return out


def global_corr_softmax_stereo(feature0, feature1):
    """matching.py:126-151: per scanline W x W correlation, targets right of x masked, disp = x - E[x']."""
    b, c, h, w = feature0.shape
    r0 = feature0.permute(0, 2, 3, 1)
    r1 = feature1.permute(0, 2, 1, 3)
    corr = torch.matmul(r0, r1) / (c ** 0.5)                          # [B, h, w, w']
    xs = torch.arange(w)
    later = xs[None, :] > xs[:, None]
    corr = corr.masked_fill(later, OOB_NEG)
    prob = torch.softmax(corr, dim=-1)
    xg = torch.arange(w, dtype=feature0.dtype)
    disp = xg.view(1, 1, w) - (prob * xg).sum(-1)
    return disp.unsqueeze(1)


def depth_corr_softmax(feature0, feature1, intrinsics, pose, inv_depth_candidates,
                       from_argmax=False, bidir=False):
    """matching.py:203-282.  intrinsics ``[B,3,3]`` already divided by the feature stride,
    pose ``[B,4,4]``, inv_depth_candidates ``[D]`` (inverse depths).  Returns inverse depth ``[B',1,h,w]``.
    """
    if bidir:
        feature0, feature1 = torch.cat([feature0, feature1], 0), torch.cat([feature1, feature0], 0)
        intrinsics = intrinsics.repeat(2, 1, 1)
        pose = torch.cat([pose, torch.inverse(pose)], 0)
    b, c, h, w = feature0.shape
    dt = feature0.dtype
    cand = inv_depth_candidates.to(dt)
    nd = cand.numel()
    grid = pixel_grid(h, w, dt)
    homog = torch.cat([grid, torch.ones(1, h, w, dtype=dt)], 0).flatten(1)        # [3, L]
    rays = torch.inverse(intrinsics) @ homog                                       # [B, 3, L]
    rays = pose[:, :3, :3] @ rays
    depth = 1.0 / cand
    pts = rays[:, :, None, :] * depth.view(1, 1, nd, 1) + pose[:, :3, 3].view(b, 3, 1, 1)
    proj = (intrinsics @ pts.reshape(b, 3, -1)).reshape(b, 3, nd, h * w)
    z = proj[:, 2].clamp(min=1e-3)
    px = proj[:, 0] / z                                                            # [B, D, L]
    py = proj[:, 1] / z
    x0 = torch.floor(px)
    y0 = torch.floor(py)
    fx, fy = px - x0, py - y0
    x0, y0 = x0.long(), y0.long()
    f1 = feature1.permute(0, 2, 3, 1)
    f0 = feature0.flatten(2).transpose(1, 2)                                       # [B, L, C]
    bi = torch.arange(b).view(b, 1, 1)
    logits = torch.zeros(b, nd, h * w, dtype=dt)
    for oy, ox, wt in ((0, 0, (1 - fx) * (1 - fy)), (0, 1, fx * (1 - fy)),
                       (1, 0, (1 - fx) * fy), (1, 1, fx * fy)):
        yy, xx = y0 + oy, x0 + ox
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        g = f1[bi, yy.clamp(0, h - 1), xx.clamp(0, w - 1)]                          # [B, D, L, C]
        d = (g * f0[:, None]).sum(-1)
        logits = logits + torch.where(ok, d * wt, torch.zeros_like(d))
    logits = logits / (c ** 0.5)
    prob = torch.softmax(logits, dim=1)
    if from_argmax:
        out = cand[prob.argmax(dim=1)]
    else:
        out = (prob * cand.view(1, nd, 1)).sum(1)
    return out.view(b, 1, h, w)


# ----------------------------------------------------------------------------------------------
# self-attention propagation
# ----------------------------------------------------------------------------------------------
def prop_global(feature0, flow, p, prefix='feature_flow_attn.'):
    """attention.py:184-215.  NOTE the reference quirk: key = k_proj(q_proj(x)), not k_proj(x)."""
    b, c, h, w = feature0.shape
    x = feature0.flatten(2).transpose(1, 2)
    q = x @ p[prefix + 'q_proj.weight'].t() + p[prefix + 'q_proj.bias']
    k = q @ p[prefix + 'k_proj.weight'].t() + p[prefix + 'k_proj.bias']
    val = flow.flatten(2).transpose(1, 2)
    prob = torch.softmax(torch.bmm(q, k.transpose(1, 2)) / (c ** 0.5), dim=-1)
    out = torch.bmm(prob, val)
    return out.transpose(1, 2).reshape(b, -1, h, w)


def prop_local(feature0, flow, p, radius, prefix='feature_flow_attn.'):
    """attention.py:217-253.  key = k_proj(x); out-of-image neighbours have key 0 (logit 0) and
    value 0 and DO take part in the softmax (zero padding of F.unfold)."""
    b, c, h, w = feature0.shape
    x = feature0.flatten(2).transpose(1, 2)
    q = (x @ p[prefix + 'q_proj.weight'].t() + p[prefix + 'q_proj.bias']).transpose(1, 2).reshape(b, c, h, w)
    k = (x @ p[prefix + 'k_proj.weight'].t() + p[prefix + 'k_proj.bias']).transpose(1, 2).reshape(b, c, h, w)
    logits, vals = [], []
    for dy in range(-radius, radius + 1):
        for dx in range(-radius, radius + 1):
            ks, _ = _shifted(k, dy, dx)
            vs, _ = _shifted(flow, dy, dx)
            logits.append((q * ks).sum(1) / (c ** 0.5))
            vals.append(vs)
    prob = torch.softmax(torch.stack(logits, 1), dim=1)                            # [B, T, h, w]
    vals = torch.stack(vals, 1)                                                    # [B, T, V, h, w]
    return (prob.unsqueeze(2) * vals).sum(1)
