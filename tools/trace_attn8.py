"""Diagnostic: section-level cycle stamps of window_attn8_kernel (needs a -DUM_TRACE build:
``python -m unimatch_amd.build --variant trace8 --only window_attn.hip -DUM_TRACE``, then ``UM_LIB=unimatch_amd/_variants/libtrace8.so``).
Config-2 geometry at batch 8 (16 streams, 64 x 96 map, 32 x 48 windows = 768 workgroups, 3 rounds of 256 CUs) through
um_window_attn_qproj_merge_fwd; lane 0 of waves 0 and 4 (group A / group B) of every 37th workgroup stamps s_memtime at the section
boundaries of its first 24 periods.  Sections -- group A: prepare | QK^T | bias + softmax | PV | wait | barrier;  group B: prepare |
softmax | PV (+ LDS-DMA requests) | QK^T + bias | wait | barrier."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimatch_amd import _abi
from unimatch_amd.ops import HipOps
ops = HipOps('exact'); lib = _abi.load()
s_, h, w, c = 16, 64, 96, 128
g = torch.Generator(device='cuda').manual_seed(0)
norm = torch.nn.LayerNorm(c).cuda()
wq, wk, wv, wm = (torch.randn(c, c, device='cuda', generator=g) * 0.09 for _ in range(4))
m = s_ * h * w
x = torch.randn(m, c, device='cuda', generator=g) * 1.5
kv, _, n2 = ops.linear_planes(x, (wk, wv))
fn = lambda: ops.window_attention_qproj_merge(x, wq, (kv, m, n2, 0), (kv, m, n2, c), s_, h, w, 32, 48, 0, 0, s_ // 2, wm, norm, x)
nwg = 768
buf = torch.zeros((nwg // 37 + 1) * 2 * (24 * 8 + 8), dtype=torch.int64, device='cuda')
raw = ctypes.CDLL(_abi.LIB_PATH)
for _ in range(2):
    fn()
torch.cuda.synchronize()
raw.um_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
fn()
torch.cuda.synchronize()
raw.um_debug_set_trace(ctypes.c_void_p(0))
b = buf.cpu().view(-1, 24 * 8 + 8)
names = {0: ['prep', 'QK', 'softmax', 'PV', 'wait', 'barrier'], 1: ['prep', 'softmax', 'PV', 'QK', 'wait', 'barrier']}
t0 = b[:, 24 * 8][b[:, 24 * 8] > 0].min().item()
for i in range(b.shape[0]):
    st = b[i, :24 * 8].view(24, 8)
    if st[2, 0] == 0:
        continue
    d = (st[:, 1:7] - st[:, 0:6]).double()
    per = (st[1:, 0] - st[:-1, 0]).double()
    grp = i % 2
    print(f'wg {(i // 2) * 37:4d} group {"AB"[grp]} start {b[i, 192].item() - t0:9d} loop {b[i, 193].item() - b[i, 192].item():8d} '
          f'total {(b[i, 194].item() - b[i, 192].item()) if grp == 0 else 0:8d} cyc  per-period {per[2:22].mean().item():7.0f}  ' +
          '  '.join(f'{n} {d[2:22, j].mean().item():6.0f}' for j, n in enumerate(names[grp])))
