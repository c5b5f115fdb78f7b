"""Diagnostic (GPU box): per-stage error of the product forward against an fp64 evaluation of the oracle, next to the
fp32 CPU oracle's own error -- names the first stage (encoder, Transformer, matching, propagation, refinement iteration)
where the GPU path adds error beyond the fp32 floor.

    python tests/diagnostics/stage_error.py [H W]                                  (GMFlow scale-1, exact and fast)
    python tests/diagnostics/stage_error.py --config gmstereo_s2_rr3 --size 384 1248 [--weights random|damped|conditioned]
                                            [--variants default,miopen]           (op variants of tests/diagnostics/odd_sizes.py)
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as om  # noqa: E402
from unimatch_amd import UniMatch  # noqa: E402
from unimatch_amd.ops import HipOps  # noqa: E402
from unimatch_amd.synth import CONDITIONED, CONFIGS, synth_camera, synth_images, synth_state_dict  # noqa: E402

VARIANTS = {'default': ('exact', {}), 'fast': ('fast', {}), 'miopen': ('exact', {'fused_conv': False}),
            'ffn2': ('exact', {'fused_ffn': False}), 'merge2': ('exact', {'fused_merge': False})}
WEIGHTS = {'random': {}, 'damped': dict(refine_gain=0.02), 'conditioned': CONDITIONED}


def err(a, b):
    d = (a.double() - b.double()).abs()
    return d.mean().item(), d.max().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('hw', nargs='*', type=int)
    ap.add_argument('--config', default='gmflow_s1')
    ap.add_argument('--size', nargs=2, type=int, default=None)
    ap.add_argument('--weights', default='random', choices=list(WEIGHTS))
    ap.add_argument('--variants', default='default,fast')
    ap.add_argument('--seed', type=int, default=1000)
    a = ap.parse_args()
    H, W = a.size if a.size else (tuple(a.hw) if len(a.hw) == 2 else (512, 768))
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    ck, fk = CONFIGS[a.config]
    model = UniMatch(**ck).eval()
    sd = synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, **WEIGHTS[a.weights])
    model.load_state_dict(sd)
    model = model.cuda()
    i0, i1 = synth_images(1, H, W, seed=a.seed, kind='shift', normalized=(fk['task'] != 'flow'))
    kw = dict(fk)
    if fk['task'] == 'depth':
        k, pose = synth_camera(1, H, W)
        kw.update(intrinsics=k, pose=pose)
    okw = dict(kw, num_scales=ck['num_scales'], upsample_factor=ck['upsample_factor'], reg_refine=ck['reg_refine'])
    to64 = lambda d: {k: (v.double() if torch.is_tensor(v) else v) for k, v in d.items()}
    t32, t64 = {}, {}
    o32 = om.unimatch_forward(sd, i0, i1, taps=t32, **okw)
    o64 = om.unimatch_forward(sd, i0.double(), i1.double(), taps=t64, **to64(okw))
    gkw = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
    epe = lambda x, y: ((x.double() - y.double()).pow(2).sum(1).sqrt() if x.dim() == 4 else (x.double() - y.double()).abs()).mean().item()
    for tag in a.variants.split(','):
        prec, attrs = VARIANTS[tag]
        ops = HipOps(prec)
        for k, v in attrs.items():
            setattr(ops, k, v)
        model.bind_ops(ops)
        model.debug_taps = tg = {}
        og = model(i0.cuda(), i1.cuda(), **gkw)['flow_preds'][0].cpu()
        print(f'--- {a.config} 1x{H}x{W}  weights={a.weights}  variant={tag} (precision={prec})')
        first = None
        for k in t64:
            if k not in tg:
                continue
            gm, gx = err(tg[k].cpu(), t64[k])
            cm, cx = err(t32[k], t64[k])
            flag = ''
            if gm > 2.0 * cm + 1e-7:
                flag = '  <-- GPU error above 2x the fp32 port'
                first = first or k
            print(f'{k:16s} |x| {t64[k].abs().mean():9.3f}   GPU-vs-fp64 mean {gm:.2e} max {gx:.2e}   '
                  f'CPUfp32-vs-fp64 mean {cm:.2e} max {cx:.2e}   ratio {gm / max(cm, 1e-30):6.2f}{flag}')
        print(f'final EPE: GPU vs fp64 {epe(og, o64):.3e}   CPU fp32 vs fp64 {epe(o32, o64):.3e}   GPU vs CPU fp32 {epe(og, o32):.3e}'
              f'   first stage above 2x the port: {first}', flush=True)


if __name__ == '__main__':
    main()
