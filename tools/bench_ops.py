"""Micro-benchmark of the HIP entry points at BASELINE sizes (GPU box).  Kernel time comes from the library's
own hipEvents (um_timing_*), so split/convert pre-passes are reported separately from the main kernels.

    python tools/bench_ops.py [attn] [gsv] [local] [linear] [conv] [--iters N] [--precision exact|fast] [--quick]
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimatch_amd import _abi  # noqa: E402
from unimatch_amd.ops import HipOps  # noqa: E402

KNAMES = ['window_attn', 'gsv', 'split_planes', 'local_corr', 'cost_volume', 'prop_local', 'depth_corr', 'linear',
          'instance_norm', 'convex_upsample', 'ffn', 'conv']
PEAK = 2.5e15
PEAK_HBM = 8.0e12           # B/s, MI355X_MICROARCH.md
PEAK_VALU_F32 = 157.3e12    # FLOP/s fp32 vector FMA, MI355X_MICROARCH.md


def collect(lib):
    out = {}
    for kid, name in enumerate(KNAMES):
        ms, n = ctypes.c_double(0), ctypes.c_int(0)
        lib.um_timing_collect(kid, ctypes.byref(ms), ctypes.byref(n))
        if n.value:
            out[name] = (ms.value / n.value, n.value)
    return out


def run(label, fn, flops, lib, iters, kernel, issued=1.0, min_bytes=None):
    """``min_bytes``: the kernel is a gather / short-reduction kernel (K3, K4, K6, K7) -- its rooflines are HBM
    (compulsory bytes / time against 8 TB/s) and fp32 VALU (algorithmic FLOPs / time against 157 TF/s), not the matrix pipe."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    lib.um_timing_enable(-1)
    collect(lib)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    lib.um_timing_enable(0)
    t = collect(lib)
    ms = t[kernel][0]
    extra = ' '.join(f'{k}={v[0] * v[1] / iters:.3f}ms' for k, v in t.items() if k != kernel)
    tf = flops / (ms * 1e-3) / 1e12
    if min_bytes is not None:
        gbs = min_bytes / (ms * 1e-3)
        print(f'{label:46s} {ms:8.4f} ms  HBM {gbs / 1e9:8.1f} GB/s of {min_bytes / 1e6:7.1f} MB compulsory ({100 * gbs / PEAK_HBM:5.2f}% of 8 TB/s)  '
              f'VALU {tf:6.2f} TF/s fp32 ({100 * tf * 1e12 / PEAK_VALU_F32:5.2f}% of 157 TF/s)  [{extra}]', flush=True)
        return
    print(f'{label:46s} {ms:8.4f} ms  {tf:8.1f} TF/s alg ({100 * tf * 1e12 / PEAK:5.2f}% peak, issued {100 * tf * issued * 1e12 / PEAK:5.2f}%)  [{extra}]',
          flush=True)


def main():
    args = sys.argv[1:]
    iters = int(args[args.index('--iters') + 1]) if '--iters' in args else 10
    prec = args[args.index('--precision') + 1] if '--precision' in args else 'exact'
    what = [a for a in args if a in ('attn', 'gsv', 'local', 'linear', 'conv')] or ['attn', 'gsv', 'local', 'linear', 'conv']
    ops = HipOps(prec)
    lib = _abi.load()
    issued = 3.0 if prec == 'exact' else 1.0
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    C = 128
    if 'attn' in what:
        # config 2: 2B = 16 streams, 64x96 map, 2x2 windows of 32x48 = 1536 tokens
        S, h, w = 16, 64, 96
        q, k, v = (torch.randn(S, h * w, C, device=dev, generator=g) * 2 for _ in range(3))
        for sh, sw, tag in ((0, 0, 'plain'), (16, 24, 'shifted')):
            run(f'attn cfg2 S=16 64x96 win32x48 {tag}', lambda: ops.window_attention(q, k, v, h, w, 32, 48, sh, sw),
                4.0 * S * h * w * 1536 * C, lib, iters, 'window_attn', issued)
        if '--quick' not in args:
            # config 4 scale 1 (4 pairs/GPU): 8 streams, 128x192 map, 8x8 windows of 16x24 = 384 tokens
            S, h, w = 8, 128, 192
            q, k, v = (torch.randn(S, h * w, C, device=dev, generator=g) * 2 for _ in range(3))
            run('attn cfg4-s1 S=8 128x192 win16x24 shifted', lambda: ops.window_attention(q, k, v, h, w, 16, 24, 8, 12),
                4.0 * S * h * w * 384 * C, lib, iters, 'window_attn', issued)
            # config 3 scale 1 cross attention: 1-D windows of 30 on 128x240
            S, h, w = 8, 128, 240
            q, k, v = (torch.randn(S, h * w, C, device=dev, generator=g) * 2 for _ in range(3))
            run('attn cfg3-s1 S=8 128x240 1-D win30 shifted', lambda: ops.window_attention(q, k, v, h, w, 1, 30, 0, 15),
                4.0 * S * h * w * 30 * C, lib, iters, 'window_attn', issued)
    if 'linear' in what:
        # the four GEMM shapes of one Transformer block at config 2: M = 2 streams x 8 pairs x 6144 tokens
        M = 2 * 8 * 6144
        x, y = (torch.randn(M, C, device=dev, generator=g) for _ in range(2))
        wq, wk, wv, wm = (torch.randn(C, C, device=dev, generator=g) * 0.09 for _ in range(4))
        w1 = torch.randn(8 * C, 2 * C, device=dev, generator=g) * 0.06
        w2 = torch.randn(C, 8 * C, device=dev, generator=g) * 0.03
        norm = torch.nn.LayerNorm(C).to(dev)
        run('linear qkv   f32[M,128] -> planes[M,384]', lambda: ops.linear_planes(x, (wq, wk, wv)),
            2.0 * M * C * 3 * C, lib, iters, 'linear', issued)
        run('linear merge f32[M,128] -> LN f32[M,128]', lambda: ops.linear_ln(x, (wm,), norm),
            2.0 * M * C * C, lib, iters, 'linear', issued)
        run('linear ffn1  cat f32[M,256] -> gelu planes[M,1024]', lambda: ops.linear_planes(x, (w1,), a1=y, gelu=True),
            2.0 * M * 2 * C * 8 * C, lib, iters, 'linear', issued)
        hid, _, _ = ops.linear_planes(x, (w1,), a1=y, gelu=True)
        run('linear ffn2  planes[M,1024] -> LN+res f32[M,128]',
            lambda: ops.linear_ln(hid, (w2,), norm, residual=x, a_planes_k=8 * C),
            2.0 * M * 8 * C * C, lib, iters, 'linear', issued)
        run('ffn fused    f32[M,128]x2 -> LN+res f32[M,128]', lambda: ops.ffn_ln(x, y, w1, w2, norm),
            2.0 * M * 8 * C * 3 * C, lib, iters, 'ffn', issued)
    if 'conv' in what:
        # encoder convolutions at config 2 (16 images): implicit GEMM on planes vs MIOpen fp32 on the same tensors
        import time
        for (cin, cout, hh, ww, st, tag) in ((64, 64, 256, 384, 1, 'layer1 3x3 64->64 @256x384'),
                                             (64, 96, 256, 384, 2, 'layer2 3x3/2 64->96'),
                                             (96, 96, 128, 192, 1, 'layer2 3x3 96->96 @128x192'),
                                             (96, 128, 128, 192, 2, 'layer3 3x3/2 96->128'),
                                             (128, 128, 64, 96, 1, 'layer3 3x3 128->128 @64x96')):
            nb = 16
            xin = torch.randn(nb, cin, hh, ww, device=dev, generator=g)
            wt = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * 0.05
            planes, _ = ops.nchw_to_nhwc(xin, want_planes=True, want_f32=False)
            ho, wo = (hh - 1) // st + 1, (ww - 1) // st + 1
            fl = 2.0 * nb * ho * wo * cout * 9 * cin
            run(f'conv {tag}', lambda: ops.conv2d_nhwc((planes, nb, hh, ww, cin), wt, None, st, (1, 1)), fl, lib, iters,
                'conv', issued)
            for _ in range(3):
                torch.nn.functional.conv2d(xin, wt, None, stride=st, padding=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                torch.nn.functional.conv2d(xin, wt, None, stride=st, padding=1)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / iters * 1e3
            print(f'     MIOpen fp32 same conv                       {ms:8.4f} ms     {fl / ms / 1e9:8.1f} TF/s', flush=True)
    if 'conv' in what:
        # refinement-block convolutions at config 4 (4 pairs @ 1/4 resolution = 128 x 192)
        nb, hh, ww = 4, 128, 192
        rows = nb * hh * ww
        for (cin, cout, ks, pad, tag) in ((384, 256, (1, 5), (0, 2), 'GRU z|r 1x5 384->256'), (384, 128, (1, 5), (0, 2), 'GRU q 1x5 384->128'),
                                          (384, 256, (5, 1), (2, 0), 'GRU z|r 5x1 384->256'), (256, 192, (3, 3), (1, 1), 'convc2 3x3 256->192'),
                                          (256, 128, (3, 3), (1, 1), 'motion 3x3 256->128'), (128, 256, (3, 3), (1, 1), 'flow head 3x3 128->256')):
            src = ops.planes_buffer(rows, cin)
            ops.nhwc_gate(0, torch.randn(rows, cin, device=dev, generator=g), src, cin, 0, rows, cin)
            wt = torch.randn(cout, cin, ks[0], ks[1], device=dev, generator=g) * 0.03
            wb = (ops.conv_weight_planes_from(wt), None)
            out = torch.empty((rows, cout), dtype=torch.float32, device=dev)
            fl = 2.0 * rows * cout * ks[0] * ks[1] * cin
            run(f'conv {tag} @128x192 x4', lambda: ops.conv_ex((src, cin, 0, cin), (nb, hh, ww), wb, ks, 1, pad, 1, out=(out, cout, 0)),
                fl, lib, iters, 'conv', issued)
    if 'gsv' in what:
        B, h, w = 8, 64, 96
        L = h * w
        f0, f1 = (torch.randn(B, L, C, device=dev, generator=g) * 3 for _ in range(2))
        if '--zeros' in args:          # no operand toggling: what the same instruction stream does when it is not power-limited
            f0, f1 = torch.zeros_like(f0), torch.zeros_like(f1)
        run('global corr flow cfg2 B=8 L=6144', lambda: ops.global_corr_softmax_flow(f0, f1, h, w),
            B * (2.0 * L * L * C + 4.0 * L * L), lib, iters, 'gsv', issued)
        val = torch.randn(B, 2, h, w, device=dev, generator=g)
        run('global propagation cfg2 B=8 L=6144', lambda: ops.prop_global(f0, f1, val, h, w),
            B * (2.0 * L * L * C + 4.0 * L * L), lib, iters, 'gsv', issued)
        if '--quick' not in args:
            # config 5 (depth): 16 samples of 60x80 = 4800 tokens, one value channel; config 1: one sample of 40x56
            for tag, (B5, h5, w5, vch) in (('cfg5 B=16 L=4800 (depth)', (16, 60, 80, 1)), ('cfg1 B=1 L=2240', (1, 40, 56, 2))):
                L5 = h5 * w5
                q5, k5 = (torch.randn(B5, L5, C, device=dev, generator=g) * 3 for _ in range(2))
                v5 = torch.randn(B5, vch, h5, w5, device=dev, generator=g)
                run(f'global propagation {tag}', lambda: ops.prop_global(q5, k5, v5, h5, w5),
                    B5 * (2.0 * L5 * L5 * C + 2.0 * vch * L5 * L5), lib, iters, 'gsv', issued)
    if 'local' in what:
        B, h, w = 4, 128, 192
        L = h * w
        f0, f1 = (torch.randn(B, L, C, device=dev, generator=g) for _ in range(2))
        flow = torch.randn(B, 2, h, w, device=dev, generator=g) * 3
        # compulsory traffic: both feature maps once (fp32) + the flow + the output
        k4_bytes = B * (2.0 * L * C * 4 + 8.0 * L + 4.0 * 81 * L)
        k3_bytes = B * (2.0 * L * C * 4 + 8.0 * L)
        k6_bytes = B * (2.0 * L * C * 4 + 2 * 8.0 * L)
        run('cost volume cfg4 B=4 128x192 r=4 (incoherent)', lambda: ops.local_corr_with_flow(f0, f1, flow, h, w, 4),
            2.0 * B * L * 100 * C, lib, iters, 'cost_volume', min_bytes=k4_bytes)
        yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing='ij')
        smooth = torch.stack([6.3 * torch.sin(yy / 23.0) + 0.01 * xx, 4.1 * torch.cos(xx / 31.0)], 0)[None].repeat(B, 1, 1, 1).contiguous()
        run('cost volume, smooth flow (MFMA tiles)', lambda: ops.local_corr_with_flow(f0, f1, smooth, h, w, 4),
            2.0 * B * L * 100 * C, lib, iters, 'cost_volume', min_bytes=k4_bytes)
        ops.k4_mfma = False
        run('cost volume VALU kernel, incoherent flow', lambda: ops.local_corr_with_flow(f0, f1, flow, h, w, 4),
            2.0 * B * L * 100 * C, lib, iters, 'cost_volume', min_bytes=k4_bytes)
        run('cost volume VALU kernel, smooth flow', lambda: ops.local_corr_with_flow(f0, f1, smooth, h, w, 4),
            2.0 * B * L * 100 * C, lib, iters, 'cost_volume', min_bytes=k4_bytes)
        ops.k4_mfma = True
        run('local corr softmax cfg4 B=4 128x192 r=4', lambda: ops.local_corr_softmax(f0, f1, h, w, 4),
            2.0 * B * L * 81 * C, lib, iters, 'local_corr', min_bytes=k3_bytes)
        # config 3 scale 1: 1-D local correlation (9 taps) on 128 x 240
        h3, w3 = 128, 240
        s0, s1 = (torch.randn(B, h3 * w3, C, device=dev, generator=g) for _ in range(2))
        run('local corr softmax 1-D cfg3 B=4 128x240 r=4', lambda: ops.local_corr_softmax(s0, s1, h3, w3, 4, one_d=True),
            2.0 * B * h3 * w3 * 9 * C, lib, iters, 'local_corr', min_bytes=B * (2.0 * h3 * w3 * C * 4 + 4.0 * h3 * w3))
        run('prop local cfg4 B=4 128x192 r=1', lambda: ops.prop_local(f0, f1, flow, h, w, 1),
            2.0 * B * L * 9 * C + 2.0 * B * L * 9 * 2, lib, iters, 'prop_local', min_bytes=k6_bytes)
        # config 5: plane-sweep depth correlation, 16 samples of 60x80, 64 inverse-depth candidates, a sideways camera move
        B5, h5, w5, D = 16, 60, 80, 64
        g0, g1 = (torch.randn(B5, h5 * w5, C, device=dev, generator=g) for _ in range(2))
        fx = 0.9 * w5
        K = torch.tensor([[fx, 0, w5 / 2], [0, fx, h5 / 2], [0, 0, 1.0]])
        cam1 = torch.cat([torch.linalg.inv(K).reshape(-1), torch.eye(3).reshape(-1), torch.tensor([0.12, 0.02, 0.01]), K.reshape(-1)])
        cam = cam1[None].repeat(B5, 1).contiguous().to(dev)
        cand = torch.linspace(1 / 10.0, 1 / 0.5, D, device=dev)
        run('depth corr softmax cfg5 B=16 60x80 D=64', lambda: ops.depth_corr_softmax(g0, g1, h5, w5, cam, cand),
            2.0 * B5 * h5 * w5 * D * C + 8.0 * B5 * h5 * w5 * D * C, lib, iters, 'depth_corr',
            min_bytes=B5 * (2.0 * h5 * w5 * C * 4 + 4.0 * h5 * w5))


if __name__ == '__main__':
    main()
