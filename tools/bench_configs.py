"""Run the five BASELINE.json configs at their full sizes on one GPU (diagnostic; bench.py measures configs[1]).

    python tools/bench_configs.py [--only 1,5] [--steps 20] [--weights damped|conditioned]

--weights conditioned: synth.CONDITIONED (soft softmaxes, small refinement steps) -- the refinement flow is then locally
coherent, as a trained model's is, and the cost volume runs on the matrix cores (csrc/local_corr_mfma.hip); with the default
random-init statistics the scale-1 flow is incoherent everywhere and every tile takes the pixel-at-a-time path.
"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from unimatch_amd import UniMatch
from unimatch_amd.synth import CONDITIONED, CONFIGS, synth_camera, synth_images, synth_state_dict
RUNS = [  # (label, config, batch, H, W)
    ('cfg1 GMFlow-s1 1x320x448', 'gmflow_s1', 1, 320, 448),
    ('cfg2 GMFlow-s1 8x512x768', 'gmflow_s1', 8, 512, 768),
    ('cfg3 GMStereo-s2-rr3 4x512x960', 'gmstereo_s2_rr3', 4, 512, 960),
    ('cfg4 GMFlow-s2-rr6 4x512x768 (per-GPU share of B=32)', 'gmflow_s2_rr6', 4, 512, 768),
    ('cfg5 GMDepth-s1 16x480x640', 'gmdepth_s1', 16, 480, 640),
    ('(6) GMFlow-s1 4x512x768', 'gmflow_s1', 4, 512, 768),
    ('(7) GMFlow-s2-rr6 8x512x768', 'gmflow_s2_rr6', 8, 512, 768),
    ('(8) GMFlow-s1 16x512x768', 'gmflow_s1', 16, 512, 768),
    ('(9) GMStereo-s1 8x512x960', 'gmstereo_s1', 8, 512, 960),
    ('(10) GMFlow-s2-rr6 32x512x768 (config 4 as written)', 'gmflow_s2_rr6', 32, 512, 768),
    ('(11) GMFlow-s2-rr6 16x512x768', 'gmflow_s2_rr6', 16, 512, 768),
    ('(12) GMFlow-s2-rr6 2x512x768', 'gmflow_s2_rr6', 2, 512, 768),
    ('(13) GMFlow-s1 6x512x768', 'gmflow_s1', 6, 512, 768),
    ('(14) GMFlow-s1 2x512x768', 'gmflow_s1', 2, 512, 768),
    ('(15) GMFlow-s1 3x512x768', 'gmflow_s1', 3, 512, 768),
    ('(16) GMDepth-s1 2x480x640', 'gmdepth_s1', 2, 480, 640),
    ('(17) GMDepth-s1 4x480x640', 'gmdepth_s1', 4, 480, 640),
    ('(18) GMDepth-s1 8x480x640', 'gmdepth_s1', 8, 480, 640),
    ('(19) GMStereo-s2-rr3 2x512x960', 'gmstereo_s2_rr3', 2, 512, 960),
    ('(20) GMFlow-s1 4x320x448', 'gmflow_s1', 4, 320, 448),
    ('(21) GMFlow-s1 8x320x448', 'gmflow_s1', 8, 320, 448),
    ('(22) GMFlow-s1 16x320x448', 'gmflow_s1', 16, 320, 448),
    ('(23) GMFlow-s1 2x320x448', 'gmflow_s1', 2, 320, 448),
    ('(24) GMStereo-s1 2x512x960', 'gmstereo_s1', 2, 512, 960),
    ('(25) GMStereo-s1 4x512x960', 'gmstereo_s1', 4, 512, 960),
]
ARGV = sys.argv[1:]
ONLY = [int(v) for v in ARGV[ARGV.index('--only') + 1].split(',')] if '--only' in ARGV else [1, 2, 3, 4, 5]    # 6-9: extra rows of the forward_parts table
STEPS = int(ARGV[ARGV.index('--steps') + 1]) if '--steps' in ARGV else 5
WEIGHTS = ARGV[ARGV.index('--weights') + 1] if '--weights' in ARGV else 'damped'
WKW = CONDITIONED if WEIGHTS == 'conditioned' else dict(refine_gain=0.02)
K4_FLAGS = int(ARGV[ARGV.index('--k4-flags') + 1]) if '--k4-flags' in ARGV else 0     # 2: natural tiles only (no target ordering), 1: per-pixel path
REPEAT = int(ARGV[ARGV.index('--repeat') + 1]) if '--repeat' in ARGV else 1
STREAMS = int(ARGV[ARGV.index('--streams') + 1]) if '--streams' in ARGV else 0       # N >= 1: force N concurrent forwards; 0: UniMatch.forward's own plan
import ctypes
from unimatch_amd import _abi
for idx, (label, name, b, hh, ww) in enumerate(RUNS, 1):
    if idx not in ONLY:
        continue
    ck, fk = CONFIGS[name]
    model = UniMatch(**ck).eval()
    model.load_state_dict(synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, **WKW))
    model = model.cuda()
    model.ops.k4_flags = K4_FLAGS
    plain = model
    model.launch_parts = STREAMS if STREAMS >= 1 else None
    i0, i1 = synth_images(b, hh, ww, seed=3, kind='shift', normalized=(fk['task'] != 'flow'))
    kw = dict(fk)
    if fk['task'] == 'depth':
        k, pose = synth_camera(b, hh, ww)
        kw.update(intrinsics=k.cuda(), pose=pose.cuda())
    i0, i1 = i0.cuda(), i1.cuda()
    for _ in range(2):
        out = model(i0, i1, **kw)['flow_preds'][0]
    lib = _abi.load()
    for rep in range(REPEAT):
        torch.cuda.synchronize(); t = time.perf_counter()
        n = STEPS
        for _ in range(n):
            out = model(i0, i1, **kw)['flow_preds'][0]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
        # the cost volume's launches of one more forward, timed by the library's own events (UM_K_COST_VOLUME = 4)
        lib.um_timing_enable(1 << 4)
        plain.launch_parts = 1
        plain(i0, i1, **kw)
        plain.launch_parts = STREAMS if STREAMS >= 1 else None
        torch.cuda.synchronize()
        ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
        lib.um_timing_collect(4, ctypes.byref(ms), ctypes.byref(cnt))
        lib.um_timing_enable(0)
        k4 = f'  K4 {cnt.value} x {ms.value / max(cnt.value, 1):.3f} ms' if cnt.value else ''
        print(f'{label:58s} out {tuple(out.shape)} finite={bool(torch.isfinite(out).all())}  {dt*1e3:8.2f} ms/step  {b/dt:8.1f} pairs/s  peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB{k4}', flush=True)
    del model, plain, out; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
