"""The SepConvGRU gate launches of the refinement block (reg_refine.py:55-76) at config 4's geometry (4 x 128 x 192 pixels), one by one,
as refine_nhwc.NhwcUpdateBlock.iterate() issues them in the hoisted form -- and the same convolutions with the epilogue's side tensors
taken away one at a time (what the main loop alone costs, what the addend / the gate arithmetic add).

    python tools/bench_gru.py [--iters 20] [--only zr1v,q1v,zr2v,q2v]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimatch_amd.ops import HipOps  # noqa: E402

ARGV = sys.argv[1:]
ITERS = int(ARGV[ARGV.index('--iters') + 1]) if '--iters' in ARGV else 20
ONLY = ARGV[ARGV.index('--only') + 1].split(',') if '--only' in ARGV else ['zr1v', 'q1v', 'zr2v', 'q2v']
B, H, W = (int(v) for v in ARGV[ARGV.index('--geom') + 1].split('x')) if '--geom' in ARGV else (4, 128, 192)


def timed(fn, iters=ITERS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def main():
    dev = torch.device('cuda')
    ops = HipOps()
    g = torch.Generator(device=dev).manual_seed(1)
    rows = B * H * W
    geom = (B, H, W)
    G = ops.planes_buffer(rows, 512)
    ops.nhwc_gate(0, torch.randn(rows, 512, device=dev, generator=g), G, 512, 0, rows, 512)
    Hs = torch.randn(rows, 128, device=dev, generator=g)
    ZR = torch.rand(rows, 256, device=dev, generator=g)
    new = lambda c: torch.randn(rows, c, device=dev, generator=g) * 0.1
    P = {'zr1': new(256), 'q1': new(128), 'zr2': new(256), 'q2': new(128)}
    out256, out128 = torch.empty(rows, 256, device=dev), torch.empty(rows, 128, device=dev)
    wt = lambda cout, cin, ks: (ops.conv_weight_planes_from(torch.randn(cout, cin, ks[0], ks[1], device=dev, generator=g) * 0.02), None)
    cases = {   # tag: (gate, src columns, cout, ksize, pad, addend)
        'zr1v': (1, (G, 512, 256, 128), 256, (1, 5), (0, 2), P['zr1']),
        'q1v': (2, (G, 512, 256, 256), 128, (1, 5), (0, 2), P['q1']),
        'zr2v': (1, (G, 512, 128, 256), 256, (5, 1), (2, 0), P['zr2']),
        'q2v': (2, (G, 512, 256, 256), 128, (5, 1), (2, 0), P['q2']),
    }
    print(f'geometry {B} x {H} x {W} = {rows} pixels; microseconds per launch (mean of {ITERS})')
    print(f'{"launch":6s} {"K":>5s} {"mfma-min":>9s} {"as issued":>10s} {"no addend":>10s} {"plain fp32":>11s} {"plain planes":>13s}')
    for tag in ONLY:
        gate, src, cout, ks, pad, add = cases[tag]
        wb = wt(cout, src[3], ks)
        k = ks[0] * ks[1] * src[3]
        mfma_min = 3 * 2.0 * rows * cout * k / 2.5e15 * 1e6
        if gate == 1:
            full = lambda: ops.conv_gru(1, src, geom, wb, ks, pad, Hs, (G, 512, 384), z_out=ZR, addend=add)
            noadd = lambda: ops.conv_gru(1, src, geom, wb, ks, pad, Hs, (G, 512, 384), z_out=ZR)
        else:
            full = lambda: ops.conv_gru(2, src, geom, wb, ks, pad, Hs, (G, 512, 128), z=ZR, addend=add)
            noadd = lambda: ops.conv_gru(2, src, geom, wb, ks, pad, Hs, (G, 512, 128), z=ZR)
        out = out256 if cout == 256 else out128
        plain = lambda: ops.conv_ex(src, geom, wb, ks, 1, pad, 2, out=(out, cout, 0))
        planes = lambda: ops.conv_ex(src, geom, wb, ks, 1, pad, 2, outp=(G, 512, 384)) if cout == 128 else None
        t = [timed(full), timed(noadd), timed(plain), timed(planes) if cout == 128 else float('nan')]
        print(f'{tag:6s} {k:5d} {mfma_min:9.1f} {t[0]:10.1f} {t[1]:10.1f} {t[2]:11.1f} {t[3]:13.1f}', flush=True)


if __name__ == '__main__':
    main()
