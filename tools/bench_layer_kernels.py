"""Quick same-box screening of library builds (UM_LIB=unimatch_amd/_variants/libNAME.so): the two Transformer-layer kernels at
BASELINE config 2's geometry -- um_window_attn_qproj_merge_fwd (16 streams, 64 x 96 map, 32 x 48 windows; plain and shifted, with
the cross layer's kv_rotate) and um_ffn_fwd (M = 98304 tokens) -- timed by the library's own hipEvents, alternating so that both
run at the temperature they have inside the model.  Accept / reject decisions are made on tools/ab_bench.py (whole model)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimatch_amd import _abi
from unimatch_amd.ops import HipOps
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ops = HipOps('exact'); lib = _abi.load()
s_, h, w, c = 16, 64, 96, 128
g = torch.Generator(device='cuda').manual_seed(0)
norm = torch.nn.LayerNorm(c).cuda()
wq, wk, wv, wm = (torch.randn(c, c, device='cuda', generator=g) * 0.09 for _ in range(4))
m = s_ * h * w
x = torch.randn(m, c, device='cuda', generator=g) * 1.5
y = torch.randn(m, c, device='cuda', generator=g) * 1.5
w1 = torch.randn(8 * c, 2 * c, device='cuda', generator=g) * 0.06
w2 = torch.randn(c, 8 * c, device='cuda', generator=g) * 0.03
kv, _, n2 = ops.linear_planes(x, (wk, wv))
attn = lambda sh, sw, rot: ops.window_attention_qproj_merge(x, wq, (kv, m, n2, 0), (kv, m, n2, c), s_, h, w, 32, 48, sh, sw, rot, wm, norm, x)
ffn = lambda: ops.ffn_ln(x, y, w1, w2, norm)
def collect(kid):
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    lib.um_timing_collect(kid, ctypes.byref(ms), ctypes.byref(n))
    return ms.value / max(n.value, 1)
for _ in range(3):
    attn(0, 0, 0); attn(16, 24, 8); ffn()
torch.cuda.synchronize()
lib.um_timing_enable((1 << 0) | (1 << 10)); collect(0); collect(10)
for _ in range(iters):
    attn(0, 0, 0); attn(0, 0, 8); ffn(); attn(16, 24, 0); attn(16, 24, 8); ffn()
torch.cuda.synchronize()
lib.um_timing_enable(0)
a, f = collect(0), collect(10)
ref = os.environ.get('UM_LIB', 'shipped')
print(f'{os.path.basename(ref):28s} window_attn (qproj+merge) {a:.4f} ms = {4.0 * s_ * h * w * 1536 * c / a / 1e9:7.1f} TF/s alg   ffn {f:.4f} ms = {2.0 * m * 1024 * 384 / f / 1e9:7.1f} TF/s alg', flush=True)
