"""Diagnostic: eager vs HIP-graph replay at a few batch sizes (GMFlow scale-1)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from unimatch_amd import UniMatch
from unimatch_amd.graph import GraphedUniMatch
from unimatch_amd.synth import CONFIGS, synth_images, synth_state_dict
ck, fk = CONFIGS['gmflow_s1']
model = UniMatch(**ck).eval()
model.load_state_dict(synth_state_dict({k: v.shape for k, v in model.state_dict().items()}))
model = model.cuda(); graphed = GraphedUniMatch(model, clone_output=False)
def timeit(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for b, hh, ww in ((1, 320, 448), (1, 512, 768), (2, 512, 768), (8, 512, 768)):
    i0, i1 = synth_images(b, hh, ww, seed=9, kind='shift'); i0, i1 = i0.cuda(), i1.cuda()
    te = timeit(lambda: model(i0, i1, **fk), 20)
    tg = timeit(lambda: graphed(i0, i1, **fk), 20)
    print(f'B={b} {hh}x{ww}: eager {te*1e3:7.2f} ms ({b/te:7.1f} pairs/s)   graph {tg*1e3:7.2f} ms ({b/tg:7.1f} pairs/s)', flush=True)
