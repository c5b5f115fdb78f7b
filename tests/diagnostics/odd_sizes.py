"""Diagnostic (GPU box): the channels-last convolution path against the stock nn.Conv2d (MIOpen) path of the same module on
image sizes that do not tile nicely (statistics fallback, ragged tiles, row-window kernel borders), next to the spread
between two other numerically equivalent evaluations (two-launch FFN, separate merge kernel) -- the two-scale + refinement
configs are chaotic at random-init weights, so only the comparison with that spread is meaningful.
    python tests/diagnostics/odd_sizes.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from unimatch_amd import UniMatch  # noqa: E402
from unimatch_amd.ops import HipOps  # noqa: E402
from unimatch_amd.synth import CONFIGS, synth_images, synth_state_dict  # noqa: E402


def diff(a, b):
    d = a - b
    return d.pow(2).sum(1).sqrt().mean().item() if d.dim() == 4 else d.abs().mean().item()


QUICK = '--quick' in sys.argv           # two ragged cases, default vs MIOpen convolutions only
CASES = (('gmflow_s1', 1, 448, 1024), ('gmflow_s1', 2, 384, 1248), ('gmflow_s2_rr6', 1, 448, 1024),
         ('gmstereo_s2_rr3', 1, 384, 1248), ('gmflow_s2_rr6', 1, 256, 384))
VARIANTS = (('default', {}), ('miopen convs', {'fused_conv': False}), ('two-launch ffn', {'fused_ffn': False}),
            ('separate merge', {'fused_merge': False}))
if QUICK:
    CASES, VARIANTS = (CASES[1], CASES[3]), VARIANTS[:2]
if '--case' in sys.argv:                # one case, every variant
    CASES = (CASES[int(sys.argv[sys.argv.index('--case') + 1])],)
for name, b, hh, ww in CASES:
    ck, fk = CONFIGS[name]
    model = UniMatch(**ck).eval()
    model.load_state_dict(synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, refine_gain=0.02))
    model = model.cuda()
    i0, i1 = synth_images(b, hh, ww, seed=5, kind='shift', normalized=(fk['task'] != 'flow'))
    i0, i1 = i0.cuda(), i1.cuda()
    outs = {}
    for tag, attrs in VARIANTS:
        ops = HipOps('exact')
        for k, v in attrs.items():
            setattr(ops, k, v)
        model.bind_ops(ops)
        outs[tag] = model(i0, i1, **fk)['flow_preds'][0]
    ref = outs['default']
    print(f'{name:16s} {b}x{hh}x{ww}: finite={bool(torch.isfinite(ref).all())}  ' +
          '  '.join(f'|{t} - default| = {diff(o, ref):.3e}' for t, o in outs.items() if t != 'default'), flush=True)
