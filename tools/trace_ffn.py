"""Diagnostic: section-level cycle stamps of the fused FFN kernel (needs a -DUM_FFN_TRACE build:
``python -m unimatch_amd.build --variant ffntrace -DUM_FFN_TRACE``, then ``UM_LIB=unimatch_amd/_variants/libffntrace.so``).
Config-2 geometry at batch 8 (M = 98304 tokens = 768 workgroups, 3 rounds of 256 CUs) through um_ffn_fwd; lane 0 of waves 0 and 4
(role 0 / role 1 of pair 0) of every 37th workgroup stamps s_memtime at the section boundaries of its first 24 hidden slices:
  wait     s_waitcnt vmcnt(0): this wave's LDS-DMA pieces of W1(i+1), W2(i-1)
  barrier  the workgroup barrier of the slice
  issue    LDS-DMA statements of W1(i+2), W2(i), exchange reads, accumulator set-up
  stream   phase B (12 MFMAs) + phase A (24 MFMAs) with the GELU / fragment stages pinned behind them
  tail     stages no MFMA was left to hide, accumulator hand-over (send), fragment copy"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimatch_amd import _abi
from unimatch_amd.ops import HipOps
ops = HipOps('exact'); lib = _abi.load()
c = 128
g = torch.Generator(device='cuda').manual_seed(0)
m = 16 * 6144
x, y = (torch.randn(m, c, device='cuda', generator=g) * 1.5 for _ in range(2))
w1 = torch.randn(8 * c, 2 * c, device='cuda', generator=g) * 0.06
w2 = torch.randn(c, 8 * c, device='cuda', generator=g) * 0.03
norm = torch.nn.LayerNorm(c).cuda()
KV = '--kv' in sys.argv             # the FFN launch that also runs the next block's k | v projections (um_ffn_kv_fwd)
if KV:
    w4 = [torch.randn(c, c, device='cuda', generator=g) * 0.09 for _ in range(4)]
    fn = lambda: ops.ffn_ln_kv(x, y, w1, w2, norm, w4)
else:
    fn = lambda: ops.ffn_ln(x, y, w1, w2, norm)
nwg = m // 128
buf = torch.zeros((nwg // 37 + 1) * 2 * (24 * 8 + 8), dtype=torch.int64, device='cuda')
raw = ctypes.CDLL(_abi.LIB_PATH)
for _ in range(2):
    fn()
torch.cuda.synchronize()
raw.um_debug_set_ffn_trace(ctypes.c_void_p(buf.data_ptr()))
fn()
torch.cuda.synchronize()
raw.um_debug_set_ffn_trace(ctypes.c_void_p(0))
b = buf.cpu().view(-1, 24 * 8 + 8)
names = ['wait', 'barrier', 'issue', 'stream', 'tail']
t0 = b[:, 24 * 8][b[:, 24 * 8] > 0].min().item()
for i in range(b.shape[0]):
    st = b[i, :24 * 8].view(24, 8)
    if st[2, 0] == 0:
        continue
    d = (st[:, 1:6] - st[:, 0:5]).double()
    per = (st[1:, 0] - st[:-1, 0]).double()
    pro, loop = st[0, 0].item() - b[i, 192].item(), b[i, 193].item() - st[0, 0].item()
    epi = b[i, 194].item() - b[i, 193].item() if b[i, 194].item() else 0
    kv = b[i, 195].item() - b[i, 194].item() if b[i, 195].item() else 0
    print(f'wg {(i // 2) * 37:4d} role {i % 2} start {b[i, 192].item() - t0:9d} prologue {pro:6d} slices {loop:7d} epilogue {epi:6d} kv {kv:6d} ticks  per-slice {per[2:].mean().item():7.0f}  ' +
          '  '.join(f'{n} {d[2:23, j].mean().item():6.0f}' for j, n in enumerate(names)) + f'  between {(st[3:23, 0] - st[2:22, 5]).double().mean().item():5.0f}')
