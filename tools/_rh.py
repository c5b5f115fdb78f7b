import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from unimatch_amd import UniMatch
from unimatch_amd.ops import HipOps
from unimatch_amd.synth import CONFIGS, synth_images, synth_state_dict
for name, b, hh, ww in (('gmflow_s2_rr6', 4, 512, 768), ('gmstereo_s2_rr3', 4, 512, 960)):
    ck, fk = CONFIGS[name]
    model = UniMatch(**ck).eval()
    model.load_state_dict(synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, refine_gain=0.02))
    model = model.cuda()
    i0, i1 = synth_images(b, hh, ww, seed=3, kind='shift', normalized=(fk['task'] != 'flow'))
    i0, i1 = i0.cuda(), i1.cuda()
    outs = {}
    for rep in range(2):
        for hoist in (0, 1):
            HipOps.refine_hoist = bool(hoist)
            for _ in range(3):
                out = model(i0, i1, **fk)['flow_preds'][0]
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10):
                out = model(i0, i1, **fk)['flow_preds'][0]
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
            outs[hoist] = out
            print(f'{name:18s} hoist={hoist}  {dt*1e3:8.2f} ms/step  {b/dt:8.1f} pairs/s', flush=True)
    d = (outs[0] - outs[1]).abs()
    print(f'{name:18s} |hoisted - plain| mean {d.mean().item():.3e} max {d.max().item():.3e}  (|out| mean {outs[0].abs().mean().item():.3e})', flush=True)
