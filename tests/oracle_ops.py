"""Test-only hot-path backend: the same Python interface as ``unimatch_amd.ops.HipOps`` answered by the CPU
oracle.  Injected with ``model.bind_ops(OracleOps())`` so that the product's HOST logic (per-scale loop,
layout conversions, sign conventions, task dispatch) can be checked on a machine without a GPU.  It is never
used by the product."""
import torch

from oracle import hotpath as hp


def _map(tokens, h, w):
    b, _, c = tokens.shape
    return tokens.transpose(1, 2).reshape(b, c, h, w)


class OracleOps:
    def __init__(self, dtype=torch.float32):
        self.dtype = dtype
        self.calls = []

    def window_attention(self, q, k, v, h, w, win_h, win_w, shift_h=0, shift_w=0):
        self.calls.append(('window_attention', (win_h, win_w, shift_h, shift_w)))
        return hp.window_attention(q, k, v, h, w, win_h, win_w, shift_h, shift_w)

    def global_corr_softmax_flow(self, f0, f1, h, w, bidir=False):
        self.calls.append(('global_corr_softmax_flow', bidir))
        return hp.global_corr_softmax_flow(_map(f0, h, w), _map(f1, h, w), bidir)

    def global_corr_softmax_stereo(self, f0, f1, h, w):
        self.calls.append(('global_corr_softmax_stereo', None))
        return hp.global_corr_softmax_stereo(_map(f0, h, w), _map(f1, h, w))

    def local_corr_softmax(self, f0, f1, h, w, radius, one_d=False):
        self.calls.append(('local_corr_softmax', (radius, one_d)))
        return hp.local_corr_softmax(_map(f0, h, w), _map(f1, h, w), radius, one_d)

    def local_corr_with_flow(self, f0, f1, flow, h, w, radius):
        self.calls.append(('local_corr_with_flow', radius))
        return hp.local_corr_with_flow(_map(f0, h, w), _map(f1, h, w), flow, radius)

    def _softmax_value(self, q, k, value, h, w):
        b, l, c = q.shape
        prob = torch.softmax(torch.bmm(q, k.transpose(1, 2)) / (c ** 0.5), dim=-1)
        return torch.bmm(prob, value.flatten(2).transpose(1, 2)).transpose(1, 2).reshape(value.shape)

    def prop_global(self, q, k, value, h, w):
        self.calls.append(('prop_global', value.shape[1]))
        return self._softmax_value(q, k, value, h, w)

    def prop_local(self, q, k, value, h, w, radius):
        self.calls.append(('prop_local', radius))
        qm, km = _map(q, h, w), _map(k, h, w)
        c = qm.shape[1]
        logits, vals = [], []
        for dy in range(-radius, radius + 1):
            for dx in range(-radius, radius + 1):
                ks, _ = hp._shifted(km, dy, dx)
                vs, _ = hp._shifted(value, dy, dx)
                logits.append((qm * ks).sum(1) / (c ** 0.5))
                vals.append(vs)
        prob = torch.softmax(torch.stack(logits, 1), dim=1)
        return (prob.unsqueeze(2) * torch.stack(vals, 1)).sum(1)

    def depth_corr_softmax(self, f0, f1, h, w, cam, candidates, from_argmax=False):
        self.calls.append(('depth_corr_softmax', from_argmax))
        b = f0.shape[0]
        k = cam[:, 21:30].reshape(b, 3, 3)
        pose = torch.eye(4).repeat(b, 1, 1).to(cam.dtype)
        pose[:, :3, :3] = cam[:, 9:18].reshape(b, 3, 3)
        pose[:, :3, 3] = cam[:, 18:21]
        return hp.depth_corr_softmax(_map(f0, h, w), _map(f1, h, w), k, pose, candidates, from_argmax, False)
