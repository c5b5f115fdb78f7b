"""Mint golden fixtures from the REAL reference (runs only where /root/reference exists).

    python tests/golden/make_golden.py

The reference has no tests and no golden vectors of its own, so its behaviour on the
hot path is pinned here by executing it: every hot-path function (unimatch/attention.py, matching.py,
transformer.py, utils.py, position.py) and the whole ``UniMatch.forward`` for the canonical configs are run
on seeded inputs with seeded weights (``unimatch_amd.synth``); inputs that cannot be regenerated from a
seed, and all outputs, are stored as small ``.npz`` files next to this script.  The fixtures travel to the
GPU box; the reference does not.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
warnings.filterwarnings('ignore')

from unimatch import attention as ratt  # noqa: E402
from unimatch import matching as rmat  # noqa: E402
from unimatch import utils as rutil  # noqa: E402
from unimatch.position import PositionEmbeddingSine  # noqa: E402
from unimatch.transformer import FeatureTransformer  # noqa: E402
from unimatch.unimatch import UniMatch as RefUniMatch  # noqa: E402

from unimatch_amd.synth import CONFIGS, synth_camera, synth_images, synth_state_dict  # noqa: E402

torch.set_num_threads(8)
C = 128


def rnd(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB')


# ----------------------------------------------------------------------------------------------- position
def gold_position():
    out = {}
    for (h, w) in ((4, 6), (8, 12), (16, 24)):
        pe = PositionEmbeddingSine(num_pos_feats=C // 2)
        out[f'pos_{h}x{w}'] = pe(torch.zeros(1, C, h, w))[0]
    f0, f1 = rnd(1, 2, C, 8, 12), rnd(2, 2, C, 8, 12)
    a0, a1 = rutil.feature_add_position(f0, f1, 2, C)
    b0, b1 = rutil.feature_add_position(f0, f1, 1, C)
    save('position', f0=f0, f1=f1, add_k2_0=a0, add_k2_1=a1, add_k1_0=b0, add_k1_1=b1, **out)


# ----------------------------------------------------------------------------------------------- attention
def gold_attention():
    out = {}
    cases = []
    # (tag, h, w, K, shift, kind, scale)
    for scale in (1.0, 3.0):
        for shift in (False, True):
            cases.append((f'win2d_k2_s{int(shift)}_x{int(scale)}', 8, 12, 2, shift, 'win2d', scale))
    cases.append(('win2d_k4_s1_x1', 16, 24, 4, True, 'win2d', 1.0))
    cases.append(('win2d_k2_ragged_s1_x2', 10, 14, 2, True, 'win2d', 2.0))     # window 5x7 = 35 tokens
    cases.append(('winrow_k2_s0_x1', 4, 16, 2, False, 'winrow', 1.0))
    cases.append(('winrow_k4_s1_x2', 4, 24, 4, True, 'winrow', 2.0))
    cases.append(('row_x1', 5, 20, 1, False, 'row', 1.0))
    cases.append(('full_x1', 6, 10, 1, False, 'full', 1.0))
    for i, (tag, h, w, k, shift, kind, scale) in enumerate(cases):
        streams = 1 if h * w > 200 else 2
        q, kk, v = (rnd(100 + 3 * i + j, streams, h * w, C, scale=scale) for j in range(3))
        if kind == 'win2d':
            mask = rutil.generate_shift_window_attn_mask((h, w), h // k, w // k, h // k // 2, w // k // 2,
                                                         device=torch.device('cpu'))
            o = ratt.single_head_split_window_attention(q, kk, v, num_splits=k, with_shift=shift, h=h, w=w,
                                                        attn_mask=mask)
        elif kind == 'winrow':
            mask = rutil.generate_shift_window_attn_mask_1d(w, w // k, w // k // 2, device=torch.device('cpu'))
            o = ratt.single_head_split_window_attention_1d(q, kk, v, num_splits=k, with_shift=shift, h=h, w=w,
                                                           attn_mask=mask)
        elif kind == 'row':
            o = ratt.single_head_full_attention_1d(q, kk, v, h=h, w=w)
        else:
            o = ratt.single_head_full_attention(q, kk, v)
        out.update({f'{tag}.q': q, f'{tag}.k': kk, f'{tag}.v': v, f'{tag}.out': o,
                    f'{tag}.meta': np.array([h, w, k, int(shift)])})
    save('attention', **out)


# ----------------------------------------------------------------------------------------------- transformer
def gold_transformer():
    out = {}
    tr = FeatureTransformer(num_layers=6, d_model=C, nhead=1, ffn_dim_expansion=4).eval()
    sd = synth_state_dict({k: v.shape for k, v in tr.state_dict().items()}, seed=7)
    tr.load_state_dict(sd)
    for i, (attn_type, k, h, w) in enumerate((('swin', 2, 8, 12), ('swin', 1, 6, 8),
                                              ('self_swin2d_cross_1d', 2, 8, 12),
                                              ('self_swin2d_cross_swin1d', 2, 8, 12),
                                              ('self_swin2d_cross_swin1d', 4, 8, 16))):
        f0, f1 = rnd(200 + 2 * i, 2, C, h, w), rnd(201 + 2 * i, 2, C, h, w)
        with torch.no_grad():
            o0, o1 = tr(f0, f1, attn_type=attn_type, attn_num_splits=k)
        tag = f'{attn_type}_k{k}'
        out.update({f'{tag}.f0': f0, f'{tag}.f1': f1, f'{tag}.o0': o0, f'{tag}.o1': o1})
    save('transformer', **out)


# ----------------------------------------------------------------------------------------------- matching
def gold_matching():
    out = {}
    b, h, w = 2, 12, 16
    for scale, tag in ((0.5, 'soft'), (2.0, 'peaky')):
        f0, f1 = rnd(300, b, C, h, w, scale=scale), rnd(301, b, C, h, w, scale=scale)
        out[f'{tag}.f0'], out[f'{tag}.f1'] = f0, f1
        out[f'{tag}.global_flow'] = rmat.global_correlation_softmax(f0, f1, False)[0]
        out[f'{tag}.global_flow_bidir'] = rmat.global_correlation_softmax(f0, f1, True)[0]
        out[f'{tag}.local_flow_r4'] = rmat.local_correlation_softmax(f0, f1, 4)[0]
        out[f'{tag}.local_flow_r2'] = rmat.local_correlation_softmax(f0, f1, 2)[0]
        out[f'{tag}.stereo_global'] = rmat.global_correlation_softmax_stereo(f0, f1)[0]
        out[f'{tag}.stereo_local_r4'] = rmat.local_correlation_softmax_stereo(f0, f1, 4)[0]
        flow = rnd(302, b, 2, h, w, scale=3.0)
        flow[0, :, 0, 0] = torch.tensor([-40.0, 2.5])          # far outside the image
        flow[1, :, 3, 5] = torch.tensor([2.0, -1.0])           # exactly integer
        out[f'{tag}.flow_in'] = flow
        out[f'{tag}.cost_r4'] = rmat.local_correlation_with_flow(f0, f1, flow, 4)
        out[f'{tag}.cost_r2'] = rmat.local_correlation_with_flow(f0, f1, flow, 2)
        # depth
        k, pose = synth_camera(b, h * 8, w * 8)
        k = k.clone()
        k[:, :2] = k[:, :2] / 8
        cand = torch.linspace(0.1, 2.0, 64)
        cand4 = cand.view(1, 64, 1, 1).repeat(b, 1, h, w)
        out[f'{tag}.K'], out[f'{tag}.pose'], out[f'{tag}.cand'] = k, pose, cand
        out[f'{tag}.depth'] = rmat.correlation_softmax_depth(f0, f1, k, pose, cand4)[0]
        out[f'{tag}.depth_argmax'] = rmat.correlation_softmax_depth(f0, f1, k, pose, cand4, depth_from_argmax=True)[0]
        out[f'{tag}.depth_bidir'] = rmat.correlation_softmax_depth(f0, f1, k, pose, cand4, pred_bidir_depth=True)[0]
    save('matching', **out)


# ----------------------------------------------------------------------------------------------- propagation
def gold_propagation():
    out = {}
    prop = ratt.SelfAttnPropagation(in_channels=C).eval()
    sd = synth_state_dict({k: v.shape for k, v in prop.state_dict().items()}, seed=11)
    prop.load_state_dict(sd)
    b, h, w = 2, 12, 16
    f0 = rnd(400, b, C, h, w)
    out['f0'] = f0
    for vch in (2, 1):
        val = rnd(401 + vch, b, vch, h, w, scale=5.0)
        out[f'val{vch}'] = val
        with torch.no_grad():
            out[f'global{vch}'] = prop(f0, val)
            out[f'local{vch}_r1'] = prop(f0, val, local_window_attn=True, local_window_radius=1)
            out[f'local{vch}_r2'] = prop(f0, val, local_window_attn=True, local_window_radius=2)
    save('propagation', **out)


# ----------------------------------------------------------------------------------------------- end to end
E2E = {
    # config: (H, W, image kind)
    'gmflow_s1': (64, 96, 'shift'),
    'gmflow_s2_rr6': (128, 192, 'shift'),
    'gmstereo_s2_rr3': (128, 192, 'shift'),
    'gmstereo_s1': (64, 96, 'shift'),
    'gmdepth_s1': (96, 128, 'shift'),
    'gmdepth_s1_rr1': (96, 128, 'shift'),
}


REFINE_GAIN = 0.02     # see unimatch_amd.synth.synth_state_dict: untamed random refinement is chaotic


def run_reference(name, dtype, threads=8, batch=1, extra=None):
    ck, fk = CONFIGS[name]
    hh, ww, kind = E2E[name]
    model = RefUniMatch(**ck).eval()
    sd = synth_state_dict({k: v.shape for k, v in model.state_dict().items()}, refine_gain=REFINE_GAIN)
    model.load_state_dict(sd)
    model = model.to(dtype)
    i0, i1 = synth_images(batch, hh, ww, seed=1000, kind=kind, normalized=(fk['task'] != 'flow'))
    kw = dict(fk)
    if extra:
        kw.update(extra)
    if fk['task'] == 'depth':
        k, pose = synth_camera(batch, hh, ww)
        kw.update(intrinsics=k.to(dtype), pose=pose.to(dtype))
    torch.set_num_threads(threads)
    with torch.no_grad():
        out = model(i0.to(dtype), i1.to(dtype), **kw)['flow_preds']
    torch.set_num_threads(8)
    assert len(out) == 1
    return out[0]


def gold_e2e():
    out = {}
    for name in E2E:
        # the reference is not fp64-clean (coords_grid() hard-codes .float()), so its own noise floor is
        # recorded as the spread between two summation orders (8 threads vs 1 thread) of the same fp32 model
        o32 = run_reference(name, torch.float32, threads=8)
        o32_1t = run_reference(name, torch.float32, threads=1)
        out[f'{name}.fp32'] = o32
        out[f'{name}.fp32_1thread'] = o32_1t
        d1 = (o32 - o32_1t).abs()
        print(f'  {name}: out {tuple(o32.shape)} |out| mean {o32.abs().mean():.3f}  '
              f'8thr-vs-1thr mean {d1.mean():.2e} max {d1.max():.2e}')
    o = run_reference('gmflow_s1', torch.float32, extra=dict(pred_bidir_flow=True))
    out['gmflow_s1_bidir.fp32'] = o
    o = run_reference('gmdepth_s1', torch.float32, extra=dict(pred_bidir_depth=True))
    out['gmdepth_s1_bidir.fp32'] = o
    save('e2e', **out)





# ----------------------------------------------------------------------------------------------- on-disk formats
def gold_formats():
    """Bytes written by the reference's own writers (cv2 is not installed here: frame_utils only needs the import
    to succeed for writeFlow / readFlow, so an empty stub module stands in)."""
    import tempfile
    import types
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    from utils import frame_utils, file_io
    from utils.utils import InputPadder
    flow = rnd(500, 7, 9, 2, scale=20.0).numpy()
    disp = (rnd(501, 6, 5).abs() * 30).numpy().astype(np.float32)
    out = {'flow': flow, 'disp': disp}
    with tempfile.TemporaryDirectory() as d:
        frame_utils.writeFlow(os.path.join(d, 'a.flo'), flow)
        out['flo_bytes'] = np.frombuffer(open(os.path.join(d, 'a.flo'), 'rb').read(), np.uint8)
        file_io.write_pfm(os.path.join(d, 'a.pfm'), disp)
        out['pfm_bytes'] = np.frombuffer(open(os.path.join(d, 'a.pfm'), 'rb').read(), np.uint8)
    for i, (hh, ww, mode, pf) in enumerate(((436, 1024, 'sintel', 16), (375, 1242, 'kitti', 32), (64, 96, 'sintel', 8))):
        x = rnd(510 + i, 1, 3, hh, ww)
        p = InputPadder(x.shape, mode=mode, padding_factor=pf)
        y = p.pad(x)[0]
        out[f'pad{i}.meta'] = np.array([hh, ww, pf, y.shape[-2], y.shape[-1]] + list(p._pad))
        out[f'pad{i}.sum'] = np.array([y.double().sum().item()])
    save('formats', **out)


if __name__ == '__main__':
    gold_position()
    gold_attention()
    gold_transformer()
    gold_matching()
    gold_propagation()
    gold_e2e()
    gold_formats()
