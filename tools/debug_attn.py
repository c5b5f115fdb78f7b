import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hotpath as hp
from unimatch_amd.ops import HipOps
ops = HipOps('exact')
z = np.load(os.path.join(ROOT, 'tests/golden/attention.npz'))
for tag, geom in (('full_x1', None), ('win2d_k2_s0_x1', None)):
    h, w, k, shift = [int(x) for x in z[f'{tag}.meta']]
    geom = (h, w, 0, 0) if tag.startswith('full') else (h // k, w // k, 0, 0)
    q, kk, v = (torch.from_numpy(z[f'{tag}.{n}']) for n in 'qkv')
    want = hp.window_attention(q.double(), kk.double(), v.double(), h, w, *geom)
    got = ops.window_attention(q.cuda(), kk.cuda(), v.cuda(), h, w, *geom).cpu()
    nan = torch.isnan(got)
    print(tag, 'shape', tuple(got.shape), 'nan frac', nan.float().mean().item())
    print(' nan per stream', nan.float().mean(dim=(1, 2)).tolist())
    rows = nan[0].float().mean(1)
    print(' nan rows (stream 0):', [i for i in range(rows.numel()) if rows[i] > 0][:40])
    cols = nan[0].float().mean(0)
    print(' nan cols (stream 0):', [i for i in range(cols.numel()) if cols[i] > 0][:40])
    err = (got.double() - want).abs()
    err[nan] = 0
    print(' max err non-nan', err.max().item())
