"""End-to-end parity of the five BASELINE.json configs at their FULL sizes (one pair each), with the noise floor beside it.

For every config and for two weight sets -- ``random`` (reference-like init: a chaotic matcher, logits +-230) and
``conditioned`` (``synth.CONDITIONED``: soft softmaxes, fp32 agrees with fp64 to ~1e-5 px, so the north star's absolute
1e-3 px gate means something) -- prints

    EPE(GPU exact vs fp64 oracle)   EPE(fp32 port vs fp64)   EPE(fp32 port T threads vs t threads)   [EPE(GPU fast vs fp64)]

EPE = mean end-point error in pixels at full resolution (loss/flow_loss.py:24 of the reference); absolute difference for
disparity and depth.  Gates: random weights  GPU-vs-fp64 <= 1.5 x port-vs-fp64 (+1e-4);  conditioned weights  GPU-vs-fp64 < 1e-3.

The CPU legs (fp64 truth, fp32 port, thread-order noise) do not need a GPU:  ``--stage cpu``  computes them (here, in the
build container) into ``gpurun_cache/parity/`` which travels to the GPU box;  ``--stage gpu``  (default) runs the GPU legs and
computes whatever CPU leg is not cached.

    python tools/parity_fullsize.py [--stage cpu|gpu] [--configs 1,2,3,4,5] [--weights random,conditioned] [--fast] [--out file]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimatch_amd import UniMatch  # noqa: E402
from unimatch_amd.synth import CONDITIONED, CONFIGS, synth_camera, synth_images, synth_state_dict  # noqa: E402

RUNS = {   # BASELINE.json configs -> (config name, H, W); the oracle legs run one pair
    1: ('gmflow_s1', 320, 448),
    2: ('gmflow_s1', 512, 768),
    3: ('gmstereo_s2_rr3', 512, 960),
    4: ('gmflow_s2_rr6', 512, 768),
    5: ('gmdepth_s1', 480, 640),
}
CACHE = os.path.join(ROOT, 'gpurun_cache', 'parity')


def epe(a, b):
    d = a.double() - b.double()
    return (d.pow(2).sum(1).sqrt() if d.dim() == 4 else d.abs()).mean().item()


def inputs(cfg, seed=1000):
    name, hh, ww = RUNS[cfg]
    ck, fk = CONFIGS[name]
    i0, i1 = synth_images(1, hh, ww, seed=seed + cfg, kind='shift', normalized=(fk['task'] != 'flow'))
    kw = dict(fk)
    if fk['task'] == 'depth':
        k, pose = synth_camera(1, hh, ww)
        kw.update(intrinsics=k, pose=pose)
    return ck, kw, i0, i1


def weights(ck, which):
    shapes = {k: v.shape for k, v in UniMatch(**ck).state_dict().items()}
    return synth_state_dict(shapes, **(CONDITIONED if which == 'conditioned' else {}))


def cpu_legs(cfg, which, threads, low_threads, noise=True):
    """fp64 truth, fp32 port at ``threads`` and (``noise``) at ``low_threads`` (thread-order noise of the reference arithmetic)."""
    path = os.path.join(CACHE, f'cfg{cfg}_{which}.pt')
    if os.path.exists(path):
        return torch.load(path)
    from oracle import model as om
    ck, kw, i0, i1 = inputs(cfg)
    sd = weights(ck, which)
    okw = dict(kw, num_scales=ck['num_scales'], upsample_factor=ck['upsample_factor'], reg_refine=ck['reg_refine'])
    to64 = lambda d: {k: (v.double() if torch.is_tensor(v) else v) for k, v in d.items()}
    torch.set_num_threads(threads)
    t = time.time()
    o32 = om.unimatch_forward(sd, i0, i1, **okw)
    t32 = time.time() - t
    o64 = om.unimatch_forward(sd, i0.double(), i1.double(), **to64(okw))
    rec = {'o64': o64, 'o32': o32, 'threads': threads, 'low_threads': low_threads, 'port_seconds': t32}
    if not noise:
        return rec                                   # (not cached: the table wants all three legs)
    torch.set_num_threads(low_threads)
    rec['o32_low'] = om.unimatch_forward(sd, i0, i1, **okw)
    torch.set_num_threads(threads)
    os.makedirs(CACHE, exist_ok=True)
    torch.save(rec, path)
    return rec


def gpu_leg(cfg, which, precision):
    ck, kw, i0, i1 = inputs(cfg)
    model = UniMatch(**ck).eval()
    model.load_state_dict(weights(ck, which))
    model = model.cuda().set_precision(precision)
    kw = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
    out = model(i0.cuda(), i1.cuda(), **kw)['flow_preds'][0].cpu()
    del model
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stage', default='gpu', choices=['cpu', 'gpu'])
    ap.add_argument('--configs', default='1,2,3,4,5')
    ap.add_argument('--weights', default='random,conditioned')
    ap.add_argument('--fast', action='store_true', help='also run the bf16 throughput mode (never a parity claim)')
    ap.add_argument('--threads', type=int, default=min(32, os.cpu_count() or 8))
    ap.add_argument('--low-threads', type=int, default=1)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    rows = []
    hdr = (f'{"config":34s} {"weights":11s} {"GPU exact-fp64":>14s} {"port fp32-fp64":>14s} {"port T-vs-t thr":>15s} '
           f'{"GPU fast-fp64":>13s}  {"|out|":>8s}  gate')
    print(hdr, flush=True)
    for cfg in [int(c) for c in a.configs.split(',')]:
        for which in a.weights.split(','):
            rec = cpu_legs(cfg, which, a.threads, a.low_threads)
            name, hh, ww = RUNS[cfg]
            row = {'config': cfg, 'name': f'cfg{cfg} {name} 1x{hh}x{ww}', 'weights': which,
                   'port_fp32_vs_fp64': epe(rec['o32'], rec['o64']), 'port_threads_noise': epe(rec['o32'], rec['o32_low']),
                   'threads': [rec['threads'], rec['low_threads']], 'out_abs_mean': rec['o64'].abs().mean().item()}
            if a.stage == 'gpu':
                row['gpu_exact_vs_fp64'] = epe(gpu_leg(cfg, which, 'exact'), rec['o64'])
                if a.fast:
                    row['gpu_fast_vs_fp64'] = epe(gpu_leg(cfg, which, 'fast'), rec['o64'])
                if which == 'conditioned':
                    row['gate'] = 'abs<1e-3: ' + ('PASS' if row['gpu_exact_vs_fp64'] < 1e-3 else 'FAIL')
                else:
                    lim = 1.5 * row['port_fp32_vs_fp64'] + 1e-4
                    row['gate'] = '<=1.5x port: ' + ('PASS' if row['gpu_exact_vs_fp64'] <= lim else 'FAIL')
            f = lambda k: f'{row[k]:.3e}' if k in row else '-'
            print(f'{row["name"]:34s} {which:11s} {f("gpu_exact_vs_fp64"):>14s} {f("port_fp32_vs_fp64"):>14s} '
                  f'{f("port_threads_noise"):>15s} {f("gpu_fast_vs_fp64"):>13s}  {row["out_abs_mean"]:8.3f}  {row.get("gate", "")}',
                  flush=True)
            rows.append(row)
    if a.out:
        with open(a.out, 'w') as fh:
            json.dump(rows, fh, indent=1)


if __name__ == '__main__':
    main()
