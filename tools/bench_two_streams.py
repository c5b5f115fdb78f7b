"""Config 2 (GMFlow scale-1, 8 x 512x768) as ONE forward of 8 pairs against TWO concurrent forwards of 4 pairs on two HIP streams
(the samples of a batch are independent): does overlapping one half's launch tails with the other half's launches pay?

    python tools/bench_two_streams.py [--steps 20] [--parts 2]
"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from unimatch_amd import UniMatch
from unimatch_amd.synth import CONFIGS, synth_images, synth_state_dict
ARGV = sys.argv[1:]
STEPS = int(ARGV[ARGV.index('--steps') + 1]) if '--steps' in ARGV else 20
PARTS = int(ARGV[ARGV.index('--parts') + 1]) if '--parts' in ARGV else 2
STAGGER = '--stagger' in ARGV        # part k + 1 starts when part k has left its CNN encoder (Transformer of one beside the encoder of the next)
ck, fk = CONFIGS['gmflow_s1']
model = UniMatch(**ck).eval()
model.load_state_dict(synth_state_dict({k: v.shape for k, v in model.state_dict().items()}))
model = model.cuda()
model.launch_parts = 1               # round 6: UniMatch.forward would cut the batch itself (streams.forward_parts); this tool does the cutting
b = 8
i0, i1 = synth_images(b, 512, 768, seed=3, kind='shift', normalized=False)
i0, i1 = i0.cuda(), i1.cuda()
streams = [torch.cuda.Stream() for _ in range(PARTS)]
chunks = [(i0[k * b // PARTS:(k + 1) * b // PARTS].contiguous(), i1[k * b // PARTS:(k + 1) * b // PARTS].contiguous()) for k in range(PARTS)]


def whole():
    return model(i0, i1, **fk)['flow_preds'][0]


enc_done = {}


def _after_encoder(_m, _a, _o):
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    enc_done['ev'] = ev


if STAGGER:
    model.backbone.register_forward_hook(_after_encoder)


def split():
    outs = []
    cur = torch.cuda.current_stream()
    enc_done.pop('ev', None)
    for s, (a0, a1) in zip(streams, chunks):
        s.wait_stream(cur)
        if STAGGER and 'ev' in enc_done:
            s.wait_event(enc_done['ev'])
        with torch.cuda.stream(s):
            outs.append(model(a0, a1, **fk)['flow_preds'][0])
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(outs, 0)


for fn in (whole, split):
    for _ in range(3):
        out = fn()
torch.cuda.synchronize()
ref = whole()
got = split()
print('max |split - whole| =', (got - ref).abs().max().item(), ' bitwise', torch.equal(got, ref))
for rep in range(3):
    for name, fn in (('one forward of 8', whole), (f'{PARTS} concurrent forwards of {b // PARTS}', split)):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(STEPS):
            fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / STEPS
        print(f'{name:34s} {dt * 1e3:8.3f} ms/step  {b / dt:8.1f} pairs/s', flush=True)
