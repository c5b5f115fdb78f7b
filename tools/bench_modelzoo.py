"""Batch-1 whole-model latency at the resolutions of the reference's model zoo (the only performance figures the reference
publishes: MODEL_ZOO.md, one A100, fp32 PyTorch eager, mean of 100 runs after 5 warm-ups with a synchronize around every
call -- evaluate_flow.py:364-366, 401-421).  Same protocol here, random-init weights, synthetic frames, exact mode.
Context only: different hardware, and BASELINE.json's metric is batch-8 throughput (bench.py)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from unimatch_amd import UniMatch
from unimatch_amd.synth import CONFIGS, synth_camera, synth_images, synth_state_dict


def variant(name, **fwd):
    ck, fk = CONFIGS[name]
    ck, fk = dict(ck), dict(fk)
    if fwd.pop('no_refine', False):
        ck['reg_refine'] = False
        fk.pop('num_reg_refine', None)
    fk.update(fwd)
    return ck, fk


RUNS = [  # label, (ctor kwargs, forward kwargs), H, W, A100 latency published by the reference (ms)
    ('GMFlow-scale1 448x1024', variant('gmflow_s1'), 448, 1024, 26),
    ('GMFlow-scale2 448x1024', variant('gmflow_s2_rr6', no_refine=True), 448, 1024, 66),
    ('GMFlow-scale2-regrefine6 448x1024', variant('gmflow_s2_rr6'), 448, 1024, 122),
    ('GMStereo-scale1 384x1248', variant('gmstereo_s1'), 384, 1248, 23),
    ('GMStereo-scale2 384x1248', variant('gmstereo_s2_rr3', no_refine=True), 384, 1248, 58),
    ('GMStereo-scale2-regrefine3 384x1248', variant('gmstereo_s2_rr3'), 384, 1248, 86),
    ('GMDepth-scale1 480x640', variant('gmdepth_s1'), 480, 640, 17),
    ('GMDepth-scale1-regrefine1 480x640', variant('gmdepth_s1_rr1'), 480, 640, 20),
]
for label, (ck, fk), hh, ww, ref_ms in RUNS:
    model = UniMatch(**ck).eval()
    model.load_state_dict(synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, refine_gain=0.02))
    model = model.cuda()
    i0, i1 = synth_images(1, hh, ww, seed=3, kind='shift', normalized=(fk['task'] != 'flow'))
    kw = dict(fk)
    if fk['task'] == 'depth':
        k, pose = synth_camera(1, hh, ww)
        kw.update(intrinsics=k.cuda(), pose=pose.cuda())
    i0, i1 = i0.cuda(), i1.cuda()
    for _ in range(5):
        out = model(i0, i1, **kw)['flow_preds'][0]
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(100):
        torch.cuda.synchronize(); t = time.perf_counter()
        out = model(i0, i1, **kw)['flow_preds'][0]
        torch.cuda.synchronize(); tot += time.perf_counter() - t
    ms = tot / 100 * 1e3
    print(f'{label:38s} {ms:7.2f} ms per pair (batch 1, exact mode)   reference on A100 (MODEL_ZOO.md): {ref_ms:4d} ms   finite={bool(torch.isfinite(out).all())}',
          flush=True)
    del model, out; torch.cuda.empty_cache()
