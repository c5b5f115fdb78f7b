"""How the cost-volume launches of a config split their tiles between the matrix-core path and the pixel path, and how many tiles the
target-ordered mode walks (k4m_kernel's `stats` counters + the control words of the k4s_* kernels):
    python tools/k4_tile_stats.py [gmflow_s2_rr6 4 512 768] [--k4-flags 2]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from unimatch_amd import UniMatch, _abi
from unimatch_amd.synth import CONFIGS, synth_images, synth_state_dict
argv = [a for a in sys.argv[1:] if not a.startswith('--')]
name, b, hh, ww = (argv + ['gmflow_s2_rr6', '4', '512', '768'])[:4] if len(argv) >= 4 else ('gmflow_s2_rr6', '4', '512', '768')
b, hh, ww = int(b), int(hh), int(ww)
flags = int(sys.argv[sys.argv.index('--k4-flags') + 1]) if '--k4-flags' in sys.argv else 0
ck, fk = CONFIGS[name]
model = UniMatch(**ck).eval()
model.load_state_dict(synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, refine_gain=0.02))
model = model.cuda(); model.ops.k4_flags = flags
i0, i1 = synth_images(b, hh, ww, seed=3, kind='shift', normalized=(fk['task'] != 'flow'))
i0, i1 = i0.cuda(), i1.cuda()
lib = _abi.load()
real = lib.um_local_corr_with_flow_feat
log = []
def spy(*args):
    args = list(args)
    st = torch.zeros(2, dtype=torch.int32, device='cuda')
    args[-2] = ctypes.c_void_p(st.data_ptr())
    rc = real(*args)
    log.append(st)
    return rc
real_argtypes = real.argtypes
lib.um_local_corr_with_flow_feat = spy
spy.argtypes = real_argtypes
model(i0, i1, **fk)
torch.cuda.synchronize()
for i, st in enumerate(log):
    prod, pix = st.tolist()
    print(f'launch {i}: tiles on the product path {prod:6d}   on the pixel path {pix:6d}   total {prod + pix} (natural tiles: {b * (hh // 4) * (ww // 4) // 32})')
