"""Diagnostic: section-level cycle stamps of the attention layer kernel (needs a -DUM_TRACE build:
``python -m unimatch_amd.build --variant trace -DUM_TRACE [-DUM_WATTN_PIPE_DEFAULT=1]``, then ``UM_LIB=.../libtrace.so``).
Config-2 geometry at batch 8 (16 streams, 64 x 96 map, 32 x 48 windows) through um_window_attn_qproj_merge_fwd; every 37th
workgroup stamps s_memtime at the section boundaries of its first 24 key tiles."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimatch_amd import _abi
from unimatch_amd.ops import HipOps
ops = HipOps('exact'); lib = _abi.load()
args_ = [a for a in sys.argv[1:] if not a.startswith('--')]
names = args_[0].split(',') if args_ else ['s0', 's1', 's2', 's3', 's4', 's5']
s_, h, w, c = (2 if '--batch1' in sys.argv else 16), 64, 96, 128      # --batch1: the key-split small launch of one pair
g = torch.Generator(device='cuda').manual_seed(0)
norm = torch.nn.LayerNorm(c).cuda()
wq, wk, wv, wm = (torch.randn(c, c, device='cuda', generator=g) * 0.09 for _ in range(4))
m = s_ * h * w
x = torch.randn(m, c, device='cuda', generator=g) * 1.5
kv, _, n2 = ops.linear_planes(x, (wk, wv))
fn = lambda: ops.window_attention_qproj_merge(x, wq, (kv, m, n2, 0), (kv, m, n2, c), s_, h, w, 32, 48, 0, 0, s_ // 2, wm, norm, x)
nwg = 768            # (--batch1: 384 key-split workgroups; the buffer is sized for the larger grid)
buf = torch.zeros((nwg // 37 + 1) * (24 * 8 + 8), dtype=torch.int64, device='cuda')
raw = ctypes.CDLL(_abi.LIB_PATH)
for _ in range(2):
    fn()
torch.cuda.synchronize()
raw.um_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
fn()
torch.cuda.synchronize()
raw.um_debug_set_trace(ctypes.c_void_p(0))
b = buf.cpu().view(-1, 24 * 8 + 8)
t0 = b[:, 24 * 8][b[:, 24 * 8] > 0].min().item()
for i in range(b.shape[0]):
    st = b[i, :24 * 8].view(24, 8)
    if st[0, 0] == 0:
        continue
    ns = len(names)
    d = (st[:, 1:ns + 1] - st[:, 0:ns]).double()          # per-section cycles per tile
    nt = int((st[:, 0] > 0).sum().item())
    per_tile = (st[1:nt, 0] - st[:nt - 1, 0]).double().mean().item() if nt > 1 else 0.0
    print(f'wg {i*37:4d} start {b[i,192].item()-t0:9d} total {b[i,193].item()-b[i,192].item():8d} cyc  per-tile {per_tile:7.0f}  ' +
          '  '.join(f'{n} {d[1:max(nt, 2), j].mean().item():6.0f}' for j, n in enumerate(names)) + f'  tiles {nt}  first-tile-at {st[0, 0].item() - b[i, 192].item():6d}')
