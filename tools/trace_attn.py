"""Diagnostic: section-level cycle stamps of the attention kernel (needs a -DUM_TRACE build: UM_LIB=tools/abl/lib_trace.so)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimatch_amd import _abi
from unimatch_amd.ops import HipOps
ops = HipOps('exact'); lib = _abi.load()
S, h, w, C = 16, 64, 96, 128
g = torch.Generator(device='cuda').manual_seed(0)
q, k, v = (torch.randn(S, h * w, C, device='cuda', generator=g) * 2 for _ in range(3))
nwg = 768
buf = torch.zeros((nwg // 37 + 1) * (24 * 8 + 8), dtype=torch.int64, device='cuda')
raw = ctypes.CDLL(_abi.LIB_PATH)
for _ in range(2):
    ops.window_attention(q, k, v, h, w, 32, 48, 0, 0)
torch.cuda.synchronize()
raw.um_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
ops.window_attention(q, k, v, h, w, 32, 48, 0, 0)
torch.cuda.synchronize()
raw.um_debug_set_trace(ctypes.c_void_p(0))
b = buf.cpu().view(-1, 24 * 8 + 8)
names = ['bias+rescale', 'QK+dma+addr', 'bias-add', 'PV||softmax', 'dma-wait', 'barrier']
t0 = b[:, 24 * 8].min().item()
for i in range(b.shape[0]):
    st = b[i, :24 * 8].view(24, 8)
    if st[0, 0] == 0:
        continue
    d = (st[:, 1:7] - st[:, 0:6]).double()          # per-section cycles per tile
    per_tile = (st[1:, 0] - st[:-1, 0]).double().mean().item()
    print(f'wg {i*37:4d} start {b[i,192].item()-t0:8d} total {b[i,193].item()-b[i,192].item():8d} cyc  per-tile {per_tile:7.0f}  ' +
          '  '.join(f'{n} {d[2:, j].mean().item():6.0f}' for j, n in enumerate(names)))
