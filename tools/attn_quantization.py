"""Diagnostic: does the 1.5-round workgroup count of the config-2 attention launch (768 workgroups on 256 CUs x 2 resident)
cost time?  Times um_window_attn_qproj_merge_fwd at config-2 geometry for several stream counts: 16 streams = 768 workgroups
(1.5 rounds), 32 = 1536 (3 rounds), ...; perfect balance would make the time per workgroup constant."""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from unimatch_amd import _abi
from unimatch_amd.ops import HipOps
ops, lib = HipOps('exact'), _abi.load()
h, w, c = 64, 96, 128
g = torch.Generator(device='cuda').manual_seed(0)
norm = torch.nn.LayerNorm(c).cuda()
wq, wk, wv, wm = (torch.randn(c, c, device='cuda', generator=g) * 0.09 for _ in range(4))
for s_ in (8, 16, 24, 32, 48, 64):
    m = s_ * h * w
    x = torch.randn(m, c, device='cuda', generator=g) * 1.5
    kv, _, n2 = ops.linear_planes(x, (wk, wv))
    fn = lambda: ops.window_attention_qproj_merge(x, wq, (kv, m, n2, 0), (kv, m, n2, c), s_, h, w, 32, 48, 0, 0, s_ // 2, wm, norm, x)
    for _ in range(3): fn()
    torch.cuda.synchronize(); lib.um_timing_enable(1)
    ms, n = ctypes.c_double(0), ctypes.c_int(0); lib.um_timing_collect(0, ctypes.byref(ms), ctypes.byref(n))
    for _ in range(20): fn()
    torch.cuda.synchronize(); lib.um_timing_enable(0)
    lib.um_timing_collect(0, ctypes.byref(ms), ctypes.byref(n))
    t = ms.value / n.value
    wgs = s_ * 4 * 12
    print(f'streams {s_:3d}  workgroups {wgs:5d} = {wgs / 512:5.2f} rounds   {t:.4f} ms   {1e3 * t / wgs * 512:.2f} us per round-equivalent', flush=True)
