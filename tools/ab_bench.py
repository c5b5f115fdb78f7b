"""Diagnostic: same-process A/B of the full forward with individual fused kernels switched off (interleaved rounds)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from unimatch_amd import UniMatch
from unimatch_amd.ops import HipOps
from unimatch_amd.synth import CONFIGS, synth_images, synth_state_dict
ck, fk = CONFIGS['gmflow_s1']
model = UniMatch(**ck).eval()
model.load_state_dict(synth_state_dict({k: v.shape for k, v in model.state_dict().items()}))
model = model.cuda()
i0, i1 = synth_images(8, 512, 768, seed=1000, kind='shift'); i0, i1 = i0.cuda(), i1.cuda()
class Without:
    """Proxy of a HipOps that hides some methods (the model then takes its stock PyTorch path for those)."""
    def __init__(self, ops, hidden, **attrs):
        self.__dict__.update(_ops=ops, _hidden=set(hidden), **attrs)
    def __getattr__(self, k):
        if k in self._hidden:
            raise AttributeError(k)
        return getattr(self._ops, k)
base = HipOps('exact')
variants = {'all fused': base, 'torch convex upsample': Without(base, ['convex_upsample'])}
def run(ops, n=10):
    model.bind_ops(ops)
    for _ in range(2): model(i0, i1, **fk)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): model(i0, i1, **fk)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for rnd in range(3):
    print('  '.join(f'{k}: {run(v):.2f} ms' for k, v in variants.items()), flush=True)
