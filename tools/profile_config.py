"""One configuration under a profiler (tools/collect_profiles.sh: rocprofv3 --kernel-trace --stats): python tools/profile_config.py <config> <batch> <H> <W>"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from unimatch_amd import UniMatch
from unimatch_amd.synth import CONFIGS, synth_camera, synth_images, synth_state_dict
name, b, hh, ww = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ck, fk = CONFIGS[name]
model = UniMatch(**ck).eval()
model.load_state_dict(synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, refine_gain=0.02))
model = model.cuda()
model.launch_parts = 1       # one forward, launches serialised: per-kernel durations of concurrent parts overlap and do not price a kernel
i0, i1 = synth_images(b, hh, ww, seed=3, kind='shift', normalized=(fk['task'] != 'flow'))
kw = dict(fk)
if fk['task'] == 'depth':
    k, pose = synth_camera(b, hh, ww); kw.update(intrinsics=k.cuda(), pose=pose.cuda())
i0, i1 = i0.cuda(), i1.cuda()
for _ in range(4):
    model(i0, i1, **kw)
torch.cuda.synchronize()
