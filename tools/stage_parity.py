"""Stage-isolated ("teacher-forced") parity of the two-scale + refinement BASELINE configs on the REFERENCE CONSTRUCTOR's weights.

Why: with the reference constructor's weights (``torch.manual_seed(326)``) configs 3 and 4 are chaotic end to end --
the fp32 reference is tens of pixels from an fp64 evaluation of itself -- so an end-to-end EPE says nothing about any kernel.
Here the fp64 oracle (``oracle/``, pinned to the real reference by tests/golden) runs ONCE per sample with taps, and every stage
of the product is then fed the ORACLE's input of that stage (cast to fp32) and compared with the oracle's fp64 output of that
stage; the fp32 CPU port of the same stage on the same inputs is the noise floor.  No chaos amplification between stages, and
the logits stay where the reference puts them (+-230), unlike the builder-defined conditioned weights.

Stages (reference file:line):
  encoder            backbone.py:39-133 (both scales)                       images -> backbone{0,1}_s{0,1}
  up2x_s1, warp_s1   unimatch.py:160-168, geometry.py:41-72                 flow_prop_s0 -> flow_up_s1; (backbone1_s1, flow_up_s1) -> f1_warp_s1
  xfmr_s{s}          utils.py:111-131 + transformer.py:226-294              position add + all six blocks
  blk{i}_s{s}        transformer.py:42-144 (self layer + cross/FFN layer)   stream after block i-1 -> stream after block i
  match_s{s}         matching.py:7-36 / 39-83 / 126-151 / 154-200           (f0, f1) -> flow (+ up-scaled flow, stereo clamp)
  prop_s{s}          attention.py:184-253                                   (f0, flow) -> propagated flow
  k4_it{t}           matching.py:86-123                                     (ori features, flow) -> [B, 81, h, w] cost volume
  refine_it{t}       unimatch.py:295-331 + reg_refine.py:6-119              K4 + update block: flow_{t-1} -> flow_t (last: + mask)
  convex             utils.py:134-152 (unimatch.py:351)                     (flow, mask) -> full-resolution prediction
  mask_head          unimatch.py:56-58, 246-250                             (one-scale configs) (flow, f0) -> convex-combination logits
  convex1            utils.py:134-152                                       (one-scale configs) (flow, the oracle's logits) -> prediction
  upsample           unimatch.py:246-262                                    (one-scale configs) both of the above in one piece
  (depth, config 5: match_s0 = matching.py:203-250 correlation_softmax_depth on the config's intrinsics / pose)

Gate: per stage and per sample, GPU error <= 2 x the fp32 port's error (+ 4 ulp of the stage's mean
magnitude).  The GPU legs run the config's measured batch in one call (4 samples stacked); the CPU legs run per sample in a
process pool.  Nothing here needs /root/reference.

    python tools/stage_parity.py [--configs 3,4] [--kinds shift,noise] [--weights ctor326] [--seed 1000] [--blocks 1]
                                 [--workers N] [--threads T] [--cache DIR] [--stage gpu|cpu] [--out file.json]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import parity_fullsize as pf  # noqa: E402

ULP_FLOOR = 4 * 2.0 ** -24          # gate floor: 4 fp32 ulps of the stage's mean magnitude
if os.environ.get('UM_STAGE_SIZE'):  # dry runs of the harness at a reduced frame size ("H,W"; inherited by the worker processes)
    _h, _w = (int(v) for v in os.environ['UM_STAGE_SIZE'].split(','))
    pf.RUNS = {c: (r[0], _h, _w, r[3]) for c, r in pf.RUNS.items()}


# ------------------------------------------------------------------------------------------------ stage list
def stage_names(ck, kw, blocks=True):
    ns, refine = ck['num_scales'], ck['reg_refine']
    out = ['encoder']
    for s in range(ns):
        if s > 0:
            out += [f'up2x_s{s}', f'warp_s{s}']
        out.append(f'xfmr_s{s}')
        if blocks:
            out += [f'blk{i}_s{s}' for i in range(6)]
        out += [f'match_s{s}', f'prop_s{s}']
    if refine:
        for t in range(kw.get('num_reg_refine', 1)):
            out += [f'k4_it{t}', f'refine_it{t}']
        out.append('convex')
    else:
        out += ['mask_head', 'convex1', 'upsample']
    return out


def stage_outputs(name, ck, kw):
    """[(tap key, metric)]: metric 'epe' = L2 over channels (flow-like, in cells / pixels), 'abs' = absolute difference."""
    ns = ck['num_scales']
    if name == 'encoder':
        return [(f'backbone{i}_s{s}', 'abs') for s in range(ns) for i in (0, 1)]
    kind, _, tail = name.partition('_')
    if kind == 'up2x':
        return [(f'flow_up_{tail}', 'epe')]
    if kind == 'warp':
        return [(f'f1_warp_{tail}', 'abs')]
    if kind == 'xfmr':
        return [(f'f0_{tail}', 'abs'), (f'f1_{tail}', 'abs')]
    if kind.startswith('blk'):
        return [(name, 'abs')]
    if kind == 'match':
        return [(f'flow_match_{tail}', 'epe')]
    if kind == 'prop':
        return [(f'flow_prop_{tail}', 'epe')]
    if kind == 'k4':
        return [(f'cost_{tail}', 'abs')]
    if kind == 'refine':
        t = int(tail[2:])
        last = t == kw.get('num_reg_refine', 1) - 1
        return [(f'flow_{tail}', 'epe')] + ([('mask_last', 'abs')] if last else [])
    if name == 'mask_head':
        return [('up_mask', 'abs')]
    return [('pred_raw', 'epe')]              # convex / convex1 / upsample


def _err(got, truth, metric):
    """Per-sample (mean, max) of the error of ``got`` against the fp64 ``truth`` (same device), and mean |truth|."""
    d = got.double() - truth
    d = d.pow(2).sum(1).sqrt() if metric == 'epe' else d.abs()
    d = d.flatten(1)
    return d.mean(1), d.amax(1), truth.abs().flatten(1).mean(1)


def _disp(flow, task):
    return torch.cat([-flow, torch.zeros_like(flow)], 1) if task == 'stereo' else flow


def _one_scale_pad(flow, task):
    """(two-channel input of the upsampler, sign of the prediction, depth flag) of the one-scale head (unimatch.py:246-262)."""
    if task == 'stereo':
        return torch.cat([-flow, torch.zeros_like(flow)], 1), -1.0, False
    if task == 'depth':
        return torch.cat([flow, torch.zeros_like(flow)], 1), 1.0, True
    return flow, 1.0, False


def _one_scale_pred(up, sign, task, kw):
    if task == 'flow':
        return up
    up = sign * up[:, :1]
    return up.clamp(min=kw.get('min_depth', 1. / 0.5), max=kw.get('max_depth', 1. / 10)) if task == 'depth' else up


def _prev_flow(T, t, last_scale):
    return T[f'flow_prop_s{last_scale}'] if t == 0 else T[f'flow_it{t - 1}']


# ------------------------------------------------------------------------------------------------ the fp32 CPU port, stage by stage
def port_stage(name, T, p, ck, kw, img0, img1):
    """Outputs {tap key: fp32 tensor} of stage ``name`` computed by the fp32 CPU port from the ORACLE's (fp64 -> fp32) inputs."""
    from oracle import hotpath as hp
    from oracle import model as om
    f32 = lambda k: T[k].float()
    task, ns = kw['task'], ck['num_scales']
    if name == 'encoder':
        if task == 'flow':
            mean = torch.tensor(om.IMAGENET_MEAN).view(1, 3, 1, 1)
            std = torch.tensor(om.IMAGENET_STD).view(1, 3, 1, 1)
            img0, img1 = (img0 / 255. - mean) / std, (img1 / 255. - mean) / std
        feats = om.cnn_encoder(torch.cat([img0, img1], 0), p, ns)
        nb = img0.shape[0]
        return {f'backbone{i}_s{s}': (feats[s][:nb] if i == 0 else feats[s][nb:]) for s in range(ns) for i in (0, 1)}
    kind, _, tail = name.partition('_')
    s = int(tail[1:]) if tail.startswith('s') else ns - 1
    splits = kw['attn_splits_list'][s]
    tp = {k[len('transformer.'):]: v for k, v in p.items() if k.startswith('transformer.')}
    if kind == 'up2x':
        return {f'flow_up_{tail}': F.interpolate(f32(f'flow_prop_s{s - 1}'), scale_factor=2, mode='bilinear', align_corners=True) * 2}
    if kind == 'warp':
        return {f'f1_warp_{tail}': om.warp(f32(f'backbone1_{tail}'), _disp(f32(f'flow_up_{tail}'), task))}
    if kind == 'xfmr':
        f0, f1 = f32(f'backbone0_{tail}'), f32(f'f1_warp_{tail}' if s > 0 else f'backbone1_{tail}')
        f0, f1 = hp.add_position(f0, f1, splits)
        f0, f1 = hp.feature_transformer(f0, f1, tp, kw['attn_type'], splits)
        return {f'f0_{tail}': f0, f'f1_{tail}': f1}
    if kind.startswith('blk'):
        i = int(kind[3:])
        a = f32(f'xin_{tail}' if i == 0 else f'blk{i - 1}_{tail}')
        h, w = T[f'backbone0_{tail}'].shape[-2:]
        nb = a.shape[0] // 2
        return {name: hp.transformer_block(a, torch.cat([a[nb:], a[:nb]], 0), tp, i, kw['attn_type'], splits, h, w)}
    if kind == 'match':
        f0, f1, r = f32(f'f0_{tail}'), f32(f'f1_{tail}'), kw.get('corr_radius_list', (-1,) * ns)[s]
        if task == 'depth':
            k_cur = kw['intrinsics'].float().clone()
            k_cur[:, :2] = k_cur[:, :2] / (ck['upsample_factor'] * 2 ** (ns - 1 - s))
            cand = torch.linspace(kw.get('min_depth', 1. / 0.5), kw.get('max_depth', 1. / 10), kw.get('num_depth_candidates', 64))
            fp = hp.depth_corr_softmax(f0, f1, k_cur, kw['pose'].float(), cand, kw.get('depth_from_argmax', False), False)
        elif r == -1:
            fp = hp.global_corr_softmax_flow(f0, f1, False) if task == 'flow' else hp.global_corr_softmax_stereo(f0, f1)
        else:
            fp = hp.local_corr_softmax(f0, f1, r, one_d=(task == 'stereo'))
        flow = fp if s == 0 else f32(f'flow_up_{tail}') + fp
        return {f'flow_match_{tail}': flow.clamp(min=0) if task == 'stereo' else flow}
    if kind == 'prop':
        r = kw['prop_radius_list'][s]
        f0, flow = f32(f'f0_{tail}'), f32(f'flow_match_{tail}')
        return {f'flow_prop_{tail}': hp.prop_local(f0, flow, p, r) if r > 0 else hp.prop_global(f0, flow, p)}
    ls = ns - 1
    if kind in ('k4', 'refine'):
        t = int(tail[2:])
        flow = _prev_flow(T, t, ls).float()
        corr = hp.local_corr_with_flow(f32(f'backbone0_s{ls}'), f32(f'backbone1_s{ls}'), _disp(flow, task), 4)
        if kind == 'k4':
            return {f'cost_{tail}': corr}
        proj = F.conv2d(f32(f'f0_s{ls}'), p['refine_proj.weight'], p['refine_proj.bias'])
        net, mask, delta = om.update_block(torch.tanh(proj[:, :128]), torch.relu(proj[:, 128:]), corr, flow, p)
        flow = flow + delta
        out = {f'flow_{tail}': flow.clamp(min=0) if task == 'stereo' else flow}
        if t == kw.get('num_reg_refine', 1) - 1:
            out['mask_last'] = mask
        return out
    if kind == 'convex':
        flow = f32(f'flow_it{kw.get("num_reg_refine", 1) - 1}')
        return {'pred_raw': om.convex_upsample(flow, f32('mask_last'), ck['upsample_factor'])}
    # one-scale configs: upsampler head + convex upsampling (unimatch.py:246-262), as a whole and in its two pieces
    flow, f0 = f32(f'flow_prop_s{ls}'), f32(f'f0_s{ls}')
    pad, sign, is_depth = _one_scale_pad(flow, task)
    if name == 'mask_head':
        return {'up_mask': om.upsampler_mask(pad, f0, p)}
    mask = f32('up_mask') if name == 'convex1' else om.upsampler_mask(pad, f0, p)
    return {'pred_raw': _one_scale_pred(om.convex_upsample(pad, mask, ck['upsample_factor'], is_depth=is_depth), sign, task, kw)}


# ------------------------------------------------------------------------------------------------ the product, stage by stage
def _tok(fmap):
    return fmap.flatten(2).transpose(1, 2).contiguous()


def _map(tokens, h, w):
    b, _, c = tokens.shape
    return tokens.transpose(1, 2).reshape(b, c, h, w)


def gpu_stage(name, G, model, ck, kw, img0, img1):
    """Outputs {tap key: CUDA tensor} of stage ``name`` computed by the product (HipOps through the C ABI) from the oracle's
    inputs ``G`` (fp32 CUDA tensors, the whole batch stacked).  Mirrors unimatch_amd/model.py's forward piece by piece."""
    from unimatch_amd.model import _IMAGENET_MEAN, _IMAGENET_STD
    from unimatch_amd.refine_nhwc import NhwcUpdateBlock
    ops = model.ops
    task, ns = kw['task'], ck['num_scales']
    nb = img0.shape[0]
    if name == 'encoder':
        input_norm = None
        if task == 'flow':                       # as unimatch_amd/model.py: folded into the stem's image packing on the GPU
            if model.backbone.takes_raw_images(ops, img0):
                input_norm = (_IMAGENET_MEAN, _IMAGENET_STD)
            else:
                mean, std = model._constants(img0.device)
                img0, img1 = (img0 / 255. - mean) / std, (img1 / 255. - mean) / std
        feats = model.backbone(torch.cat([img0, img1], 0), ops, input_norm)[::-1]
        return {f'backbone{i}_s{s}': (feats[s][:nb] if i == 0 else feats[s][nb:]) for s in range(ns) for i in (0, 1)}
    kind, _, tail = name.partition('_')
    s = int(tail[1:]) if tail.startswith('s') else ns - 1
    splits = kw['attn_splits_list'][s]
    ls = ns - 1
    h, w = G[f'backbone0_s{s}'].shape[-2:]
    if kind == 'up2x':
        return {f'flow_up_{tail}': ops.flow_upsample2x(G[f'flow_prop_s{s - 1}'], 2.0)}
    if kind == 'warp':
        out = ops.flow_warp(_tok(G[f'backbone1_{tail}']), _disp(G[f'flow_up_{tail}'], task).contiguous(), h, w)
        return {f'f1_warp_{tail}': _map(out, h, w)}
    if kind == 'xfmr':
        pos = model._position(h, w, splits, img0.device)
        t0, t1 = _tok(G[f'backbone0_{tail}']), _tok(G[f'f1_warp_{tail}' if s > 0 else f'backbone1_{tail}'])
        t0, t1 = model.transformer(ops, t0 + pos, t1 + pos, h, w, kw['attn_type'], splits)
        return {f'f0_{tail}': _map(t0, h, w), f'f1_{tail}': _map(t1, h, w)}
    if kind.startswith('blk'):
        from unimatch_amd.model import attention_windows
        i = int(kind[3:])
        stream = G[f'xin_{tail}' if i == 0 else f'blk{i - 1}_{tail}']
        blk = model.transformer.layers[i]
        shift = ('swin' in kw['attn_type']) and splits > 1 and i % 2 == 1
        g_self = attention_windows(kw['attn_type'], True, splits, h, w, shift)
        g_cross = attention_windows(kw['attn_type'], False, splits, h, w, shift)
        a = blk.self_attn(ops, stream, stream, h, w, g_self)
        if getattr(ops, 'fused_tail', False):   # the product: keys / values of the stream as it entered the block, halves rotated
            return {name: blk.cross_attn_ffn(ops, a, stream, h, w, g_cross, kv_rotate=nb)}
        return {name: blk.cross_attn_ffn(ops, a, torch.cat([stream[nb:], stream[:nb]], 0), h, w, g_cross)}
    if kind == 'match':
        t0, t1, r = _tok(G[f'f0_{tail}']), _tok(G[f'f1_{tail}']), kw.get('corr_radius_list', (-1,) * ns)[s]
        if task == 'depth':
            up = float(ck['upsample_factor'] * 2 ** (ns - 1 - s))
            cand = model._depth_candidates(kw.get('min_depth', 1. / 0.5), kw.get('max_depth', 1. / 10), kw.get('num_depth_candidates', 64),
                                           t0.device)
            if hasattr(ops, 'depth_cam') and t0.is_cuda:
                cam = ops.depth_cam(kw['intrinsics'].to(t0.device), kw['pose'].to(t0.device), up, False)
            else:
                kc = kw['intrinsics'].float().clone()
                kc[:, :2] = kc[:, :2] / up
                pc = kw['pose'].float()
                cam = torch.cat([torch.inverse(kc).flatten(1), pc[:, :3, :3].flatten(1), pc[:, :3, 3], kc.flatten(1)], 1).contiguous()
            fp = ops.depth_corr_softmax(t0, t1, h, w, cam, cand.contiguous(), kw.get('depth_from_argmax', False))
        elif r == -1:
            fp = ops.global_corr_softmax_flow(t0, t1, h, w, False) if task == 'flow' else ops.global_corr_softmax_stereo(t0, t1, h, w)
        else:
            fp = ops.local_corr_softmax(t0, t1, h, w, r, one_d=(task == 'stereo'))
        flow = fp if s == 0 else G[f'flow_up_{tail}'] + fp
        return {f'flow_match_{tail}': flow.clamp(min=0) if task == 'stereo' else flow}
    if kind == 'prop':
        r = kw['prop_radius_list'][s]
        flow = model.feature_flow_attn(ops, _tok(G[f'f0_{tail}']), G[f'flow_match_{tail}'].contiguous(), h, w,
                                       local_window_attn=r > 0, local_window_radius=r)
        return {f'flow_prop_{tail}': flow}
    if kind in ('k4', 'refine'):
        t = int(tail[2:])
        flow = _prev_flow(G, t, ls)
        ori0, ori1 = _tok(G[f'backbone0_s{ls}']), _tok(G[f'backbone1_s{ls}'])
        disp = _disp(flow, task).contiguous()
        if kind == 'k4':
            return {f'cost_{tail}': ops.local_corr_with_flow(ori0, ori1, disp, h, w, 4)}
        last = t == kw.get('num_reg_refine', 1) - 1
        if getattr(ops, 'fused_conv', False):    # the product's path: channels-last block, K4 writes convc1's operand planes
            nhwc = NhwcUpdateBlock(ops, model.refine, model.refine_proj)
            nhwc.begin(_tok(G[f'f0_s{ls}']), nb, h, w, iterations=kw.get('num_reg_refine', 1))    # the product's (hoisted) form
            mask, delta = nhwc.iterate(ori0, ori1, disp, flow, last)
            mask = mask.reshape(nb, h, w, -1).permute(0, 3, 1, 2) if mask is not None else None
        else:                                    # injected CPU backend (harness dry run)
            proj = model.refine_proj(G[f'f0_s{ls}'])
            corr = ops.local_corr_with_flow(ori0, ori1, disp, h, w, 4)
            _, mask, delta = model.refine(torch.tanh(proj[:, :128]), torch.relu(proj[:, 128:]), corr, flow)
        flow = flow + delta
        out = {f'flow_{tail}': flow.clamp(min=0) if task == 'stereo' else flow}
        if last:
            out['mask_last'] = mask
        return out
    if kind == 'convex':
        flow = G[f'flow_it{kw.get("num_reg_refine", 1) - 1}']
        return {'pred_raw': ops.convex_upsample(flow, G['mask_last'].contiguous(), ck['upsample_factor'], False)}
    flow, f0 = G[f'flow_prop_s{ls}'], G[f'f0_s{ls}']
    pad, sign, is_depth = _one_scale_pad(flow, task)
    pad = pad.contiguous()
    if name == 'mask_head':
        mask, nhwc = model._upsample_mask(pad, f0)
        if nhwc:
            mask = mask.reshape(nb, h, w, -1).permute(0, 3, 1, 2)
        return {'up_mask': mask}
    if name == 'convex1':          # the convex combination alone, on the oracle's logits
        return {'pred_raw': _one_scale_pred(model._convex(pad, G['up_mask'].contiguous(), is_depth=is_depth), sign, task, kw)}
    return {'pred_raw': _one_scale_pred(model._upsample(pad, f0, is_depth=is_depth), sign, task, kw)}


# ------------------------------------------------------------------------------------------------ CPU legs (worker processes)
def cpu_task(task):
    """One sample: the fp64 oracle with taps (saved to ``outdir``) and the fp32 port's per-stage error against it."""
    cfg, which, kind, seed, idx, threads, outdir, blocks = task
    path = os.path.join(outdir, f'stage_cfg{cfg}_{which}_{kind}_{seed}_{idx}.pt')
    if os.path.exists(path + '.json'):
        return task[:5], json.load(open(path + '.json'))
    torch.set_num_threads(threads)
    from oracle import model as om
    ck, kw, i0, i1 = pf.case_inputs(cfg, kind, seed)
    sd = pf.weights(ck, which)
    kw = pf._slice_kw(kw, idx, idx + 1)
    i0, i1 = i0[idx:idx + 1], i1[idx:idx + 1]
    okw = dict(kw, num_scales=ck['num_scales'], upsample_factor=ck['upsample_factor'], reg_refine=ck['reg_refine'])
    t0 = time.time()
    T = {}
    with torch.no_grad():
        om.unimatch_forward(sd, i0.double(), i1.double(), taps=T,
                            **{k: (v.double() if torch.is_tensor(v) else v) for k, v in okw.items()})
        t64 = time.time() - t0
        rows = {}
        for name in stage_names(ck, kw, blocks):
            got = port_stage(name, T, sd, ck, kw, i0, i1)
            for key, metric in stage_outputs(name, ck, kw):
                mean, mx, mag = _err(_unstack_like(key, got[key], 1), _unstack_like(key, T[key], 1), metric)
                rows[f'{name}:{key}'] = {'mean': mean.item(), 'max': mx.item(), 'mag': mag.item()}
    rec = {'port': rows, 'seconds_fp64': t64, 'seconds_total': time.time() - t0}
    torch.save({k: v.contiguous() for k, v in T.items()}, path + '.tmp')
    os.replace(path + '.tmp', path)
    with open(path + '.json', 'w') as fh:
        json.dump(rec, fh)
    return task[:5], rec


class StageLegs:
    def __init__(self, workers=None, threads=None, cache=None, blocks=True):
        ncpu = os.cpu_count() or 8
        self.threads = threads or min(16, ncpu)
        self.workers = workers or max(1, min(8, ncpu // self.threads))
        self.own_dir = cache is None
        self.dir = cache or tempfile.mkdtemp(prefix='um_stage_')
        os.makedirs(self.dir, exist_ok=True)
        self.blocks = blocks
        self.pool, self.pending, self.done = None, {}, {}

    def submit(self, keys):
        import multiprocessing as mp
        if self.pool is None:
            self.pool = mp.get_context('spawn').Pool(self.workers)
        for key in keys:
            key = tuple(key)
            if key not in self.pending and key not in self.done:
                self.pending[key] = self.pool.apply_async(cpu_task, (key + (self.threads, self.dir, self.blocks),))

    def get(self, key):
        key = tuple(key)
        if key not in self.done:
            if key not in self.pending:
                self.submit([key])
            _, rec = self.pending.pop(key).get()
            self.done[key] = rec
        return self.done[key]

    def taps(self, key):
        self.get(key)
        return torch.load(os.path.join(self.dir, 'stage_cfg{}_{}_{}_{}_{}.pt'.format(*key)))

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool = None
        if self.own_dir:
            shutil.rmtree(self.dir, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ one case on the GPU
def _stack(taps, key):
    """The batch's tap: samples along dim 0; token streams ``[f0; f1]`` keep the layout [all f0; all f1]."""
    ts = [t[key] for t in taps]
    if key.startswith('xin_') or key.startswith('blk'):
        return torch.cat([t[:1] for t in ts] + [t[1:] for t in ts], 0)
    return torch.cat(ts, 0)


def _unstack_like(key, t, nb):
    """Inverse of _stack for per-sample errors: returns the tensor with dim 0 = sample (streams: f0 and f1 of a sample side by side)."""
    if key.startswith('xin_') or key.startswith('blk'):
        return torch.cat([t[:nb], t[nb:]], 1)
    return t


def run_case(legs, cfg, which, kind, seed, precision='exact', nsamples=None, backend=None):
    """All stages of one case at the config's batch -> rows [{stage, key, metric, gpu_mean[B], gpu_max[B], port_mean[B], ...}].
    ``backend``: None = the product (HipOps on the GPU); tests of the harness itself inject a CPU backend here."""
    from unimatch_amd import UniMatch
    nb = nsamples or pf.RUNS[cfg][3]
    keys = [(cfg, which, kind, seed, i) for i in range(nb)]
    legs.submit(keys)
    dev = torch.device('cuda' if backend is None else 'cpu')
    ck, kw, i0, i1 = pf.case_inputs(cfg, kind, seed)
    kw, i0, i1 = pf._slice_kw(kw, 0, nb), i0[:nb].to(dev), i1[:nb].to(dev)
    model = UniMatch(**ck).eval()
    model.load_state_dict(pf.weights(ck, which))
    model = model.to(dev)
    if backend is None:
        model.set_precision(precision)
    else:
        model.bind_ops(backend)
    recs = [legs.get(k) for k in keys]
    taps = [legs.taps(k) for k in keys]
    rows = []
    with torch.no_grad():
        need = {}

        def dev32(key):                    # the oracle's tap as the fp32 CUDA input of a stage (kept while stages need it)
            if key not in need:
                need[key] = _stack(taps, key).float().to(dev)
            return need[key]

        class Lazy(dict):
            def __missing__(self, key):
                return dev32(key)
        G = Lazy()
        for name in stage_names(ck, kw, legs.blocks):
            got = gpu_stage(name, G, model, ck, kw, i0, i1)
            for key, metric in stage_outputs(name, ck, kw):
                truth = _unstack_like(key, _stack(taps, key).to(dev), nb)
                mean, mx, mag = _err(_unstack_like(key, got[key], nb), truth, metric)
                port = [r['port'][f'{name}:{key}'] for r in recs]
                rows.append({'config': cfg, 'weights': which, 'kind': kind, 'seed': seed, 'stage': name, 'key': key, 'metric': metric,
                             'gpu_mean': mean.tolist(), 'gpu_max': mx.tolist(), 'mag': mag.tolist(),
                             'port_mean': [p['mean'] for p in port], 'port_max': [p['max'] for p in port]})
                del truth
            need.clear()
            G.clear()
    del model
    if backend is None:
        torch.cuda.empty_cache()
    return rows


def gate(row):
    """(worst per-sample ratio GPU / port, passes).  Passes when EVERY sample has gpu <= 2 x port + 4 ulp of the magnitude."""
    worst, ok = 0.0, True
    for g, p, m in zip(row['gpu_mean'], row['port_mean'], row['mag']):
        worst = max(worst, g / max(p, 1e-300))
        ok = ok and g <= 2.0 * p + ULP_FLOOR * m
    return worst, ok


def fmt_rows(rows):
    out = []
    for r in rows:
        n = len(r['gpu_mean'])
        gm, pm = sum(r['gpu_mean']) / n, sum(r['port_mean']) / n
        worst, ok = gate(r)
        out.append(f"cfg{r['config']} {r['kind']:5s} {r['stage']:12s} {r['key']:15s} {r['metric']:3s} |x| {sum(r['mag']) / n:9.3e}  "
                   f"GPU {gm:9.3e} (max {max(r['gpu_max']):8.2e})  port {pm:9.3e} (max {max(r['port_max']):8.2e})  "
                   f"ratio of means {gm / max(pm, 1e-300):5.2f}  worst sample {worst:5.2f}  {'PASS' if ok else 'FAIL'}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', default='3,4')
    ap.add_argument('--weights', default='ctor326')
    ap.add_argument('--kinds', default='shift,noise')
    ap.add_argument('--seed', type=int, default=1000)
    ap.add_argument('--samples', type=int, default=None, help='samples per case (default: the batch the config is measured at)')
    ap.add_argument('--blocks', type=int, default=1, help='0: skip the per-block Transformer stages')
    ap.add_argument('--workers', type=int, default=None)
    ap.add_argument('--threads', type=int, default=None)
    ap.add_argument('--cache', default=None)
    ap.add_argument('--stage', default='gpu', choices=['gpu', 'cpu'])
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    legs = StageLegs(a.workers, a.threads, a.cache, bool(a.blocks))
    cases = [(int(c), w, k, a.seed) for c in a.configs.split(',') for w in a.weights.split(',') for k in a.kinds.split(',')]
    nb = lambda cfg: a.samples or pf.RUNS[cfg][3]
    legs.submit([c + (i,) for c in cases for i in range(nb(c[0]))])
    print(f'# {len(cases)} cases, CPU legs on {legs.workers} workers x {legs.threads} threads, taps in {legs.dir}', flush=True)
    rows, bad = [], 0
    try:
        if a.stage == 'cpu':
            assert a.cache, '--stage cpu needs --cache'
            for c in cases:
                for i in range(nb(c[0])):
                    rec = legs.get(c + (i,))
                    print(c + (i,), f"fp64 {rec['seconds_fp64']:.0f} s, total {rec['seconds_total']:.0f} s", flush=True)
            return
        for c in cases:
            t0 = time.time()
            sub = run_case(legs, *c, nsamples=nb(c[0]))
            name, hh, ww, _ = pf.RUNS[c[0]]
            print(f'## cfg{c[0]} {name} {nb(c[0])}x{hh}x{ww}  weights={c[1]}  images={c[2]}  seed={c[3]}  ({time.time() - t0:.0f} s)', flush=True)
            for line in fmt_rows(sub):
                print(line, flush=True)
            bad += sum(not gate(r)[1] for r in sub)
            rows += sub
            if a.out:
                with open(a.out, 'w') as fh:
                    json.dump(rows, fh)
    finally:
        legs.close()
    print(f'# stages failing the 2x gate: {bad} of {len(rows)}')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
