"""Typed Python wrappers over the C ABI: one method per reference hot-path function.

``HipOps`` is the ONLY implementation of the hot path that the product ships.  Every method checks
shapes / dtypes / contiguity, allocates the output (and scratch workspace) as torch tensors, and enqueues
the HIP kernels on torch's current stream through ``ctypes``.  Feature tensors are token-major
``[N, L, C]`` fp32 (C = 128); flow-like tensors are ``[N, V, h, w]`` fp32 as in the reference.
"""
import ctypes
import weakref

import torch

from . import _abi

PRECISIONS = {'exact': _abi.MODE_EXACT, 'fast': _abi.MODE_FAST}


class KernelTimer:
    """Optional per-launch HIP-event timing on the stream the kernels are launched on."""

    def __init__(self):
        self.records = []          # (name, start_event, end_event, meta)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, s, e, meta in self.records:
            d = out.setdefault(name, {'calls': 0, 'ms': 0.0, 'meta': meta})
            d['calls'] += 1
            d['ms'] += s.elapsed_time(e)
        return out


def _ptr(t):
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check_tokens(name, t, n=None, tokens=None):
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.is_contiguous()):
        raise ValueError(f'{name}: expected a contiguous CUDA float32 [N, L, C] tensor, got '
                         f'{tuple(t.shape)} {t.dtype} {t.device} contiguous={t.is_contiguous()}')
    if n is not None and t.shape[0] != n or tokens is not None and t.shape[1] != tokens:
        raise ValueError(f'{name}: shape {tuple(t.shape)} does not match N={n}, L={tokens}')


def _check_map(name, t, n, h, w):
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.is_contiguous()
            and t.shape[0] == n and t.shape[2] == h and t.shape[3] == w):
        raise ValueError(f'{name}: expected a contiguous CUDA float32 [{n}, V, {h}, {w}] tensor, got '
                         f'{tuple(t.shape)} {t.dtype}')


def pack_kv4_weights(weights):
    """The four ``[128, 128]`` projection weights (k_self, v_self, k_cross, v_cross) packed as the ``[256, 256]`` matrix
    ``um_kv4_fwd`` / ``um_ffn_kv_fwd`` stream like a W1 slice of the FFN: row ``32 c + r`` (chunk c = 0..7, r = 0..31) carries
    ``W4[32 c + r]`` in columns 0..127 (the rows role 0 of a wave pair multiplies: k_self | v_self) and ``W4[256 + 32 c + (r ^ 16)]``
    in columns 128..255 (role 1 reads ring rows permuted by ^16: k_cross | v_cross), ``W4`` = the four weights stacked."""
    w4 = torch.cat([w.detach().float() for w in weights], 0)                         # [512, 128]
    perm = torch.arange(32, device=w4.device) ^ 16
    cross = w4[256:].view(8, 32, 128)[:, perm].reshape(256, 128)
    return torch.cat([w4[:256], cross], 1).contiguous()


class HipOps:
    """The hot path on MI355X.  ``precision``: 'exact' (fp16 hi+lo split MFMA operands) or 'fast' (bf16)."""

    fused_tail = True          # Transformer-layer linears / LayerNorm / GELU / residual run on um_linear_fwd
    fused_ffn = True           # ... and the FFN as one kernel (um_ffn_fwd) instead of two um_linear_fwd launches
    fused_merge = True         # merge + LayerNorm (+ residual) in the attention kernel's epilogue
    fused_qproj = True         # ... and the query projection in its prologue (um_window_attn_qproj_merge_fwd)
    block_kv = True            # one k | v projection launch per Transformer block (both layers' keys / values: um_kv4_fwd)
    refine_hoist = True        # refinement loop: the iteration-invariant share of the GRU gate convolutions computed once per scale
    fused_kv = True            # ... which, from block 1 on, is the previous block's FFN epilogue (um_ffn_kv_fwd): no launch at all
    # (class attributes: tests and tools/ab_bench.py flip them programmatically; the product reads no environment variable)
    fused_conv = True          # encoder convolutions + InstanceNorm in NHWC on um_conv2d_fwd / um_nhwc_instance_norm
    CONV_MODE = 0              # ... always in the exact arithmetic: 'fast' (bf16) is a property of the matching path only
    WSHIFT = 10                # weights are scaled by 2^10 before the fp16 split (exact), see linear.hip

    def __init__(self, precision='exact'):
        if precision not in PRECISIONS:
            raise ValueError(f'precision must be one of {sorted(PRECISIONS)}')
        self.lib = _abi.load()                # raises HipExtensionError when the library is missing
        if not torch.cuda.is_available():
            raise _abi.HipExtensionError('no GPU visible: the UniMatch hot path only runs on the HIP extension')
        self.precision = precision
        self.mode = PRECISIONS[precision]
        self.timer = None                     # set to a KernelTimer() to time launches with HIP events
        self.nplanes = 2 if precision == 'exact' else 1
        self._wcache = {}                     # weight planes, keyed by the identity + version of the fp32 tensors

    # ------------------------------------------------------------------ helpers
    def _ws(self, nbytes, device):
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)

    def _launch(self, name, fn, meta=None):
        if self.timer is None:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        self.timer.records.append((name, s, e, meta))
        return r

    # ------------------------------------------------------------------ attention
    def window_attention(self, q, k, v, h, w, win_h, win_w, shift_h=0, shift_w=0):
        """softmax(q k^T/sqrt(C) + shift mask) v inside windows; q, k, v ``[S, h*w, C]``."""
        s, l, c = q.shape
        _check_tokens('q', q, tokens=h * w)
        _check_tokens('k', k, s, l)
        _check_tokens('v', v, s, l)
        out = torch.empty_like(q)
        nbytes = self.lib.um_window_attn_workspace_bytes(s, l, c, self.mode)
        ws = self._ws(nbytes, q.device)
        n = win_h * win_w
        meta = {'flops': 4.0 * s * l * n * c, 'bytes': 4.0 * 4 * s * l * c}
        code = self._launch('window_attn', lambda: self.lib.um_window_attn_fwd(
            _ptr(q), _ptr(k), _ptr(v), _ptr(out), s, h, w, c, win_h, win_w, shift_h, shift_w,
            self.mode, _ptr(ws), ws.numel(), _stream()), meta)
        _abi.check(code, 'um_window_attn_fwd')
        return out

    # ------------------------------------------------------------------ fused Transformer-layer tail
    def _cache_get(self, tag, tensors):
        """Cached value for these tensor OBJECTS at their current versions.  Keyed by identity and validated through weak
        references: a ``data_ptr`` key is wrong for temporaries (the allocator hands a freed weight's address to the next
        tensor of the same shape)."""
        # identity + autograd version + storage address + device + dtype: `p.data = ...`, `.to(device)`, `.float()` keep id and
        # version but move / retype the storage; an IN-PLACE edit through `.data` (`p.data.mul_(2)`) changes none of these and
        # needs invalidate_weights() (UniMatch.invalidate_weights(), or check_weights=True for a per-forward fingerprint)
        key = (tag,) + tuple((id(t), t._version, t.data_ptr(), str(t.device), t.dtype) for t in tensors)
        hit = self._wcache.get(key)
        if hit is not None and all(r() is t for r, t in zip(hit[0], tensors)):
            return key, hit[1]
        return key, None

    def invalidate_weights(self):
        """Forget every cached operand plane (call after editing weights in place through ``.data``) and every split
        workspace (a launch that was cut short -- aborted capture, device error -- may have left arrival counters non-zero)."""
        self._wcache.clear()
        self.__dict__.pop('_split_ws', None)
        self.cache_generation = getattr(self, 'cache_generation', 0) + 1

    def _cache_put(self, key, tensors, value):
        if len(self._wcache) > 256:
            self._wcache = {k: v for k, v in self._wcache.items() if all(r() is not None for r in v[0])}
            if len(self._wcache) > 256:
                self._wcache.clear()
        self._wcache[key] = (tuple(weakref.ref(t) for t in tensors), value)
        # counts the builds of shared (stream-independent) cache entries: ConcurrentUniMatch only runs its parts on separate streams
        # when the previous forward built none -- an entry is written on the stream that first needs it
        self.cache_generation = getattr(self, 'cache_generation', 0) + 1
        return value

    def weight_planes(self, weights):
        """MFMA operand planes of ``cat(weights, 0)`` ([N, K] fp32 each, same K), cached until a weight changes."""
        key, hit = self._cache_get('lin', weights)
        if hit is not None:
            return hit
        w = (weights[0] if len(weights) == 1 else torch.cat(list(weights), 0)).detach().float().contiguous()
        n, k = w.shape
        self._check_weight_range(w, self.WSHIFT if self.mode == 0 else 0, 'Linear weight')
        planes = torch.empty(self.lib.um_planes_bytes(n, k, self.mode), dtype=torch.uint8, device=w.device)
        _abi.check(self.lib.um_weight_planes(_ptr(w), _ptr(planes), n, k, self.WSHIFT, self.mode, _stream()),
                   'um_weight_planes')
        return self._cache_put(key, weights, (planes, n, k))

    @staticmethod
    def _check_weight_range(w, shift, what):
        """Exact mode stores weights as fp16 hi + lo planes of ``w * 2^shift``: anything at or above 65504 / 2^shift would become
        inf in the hi plane (and NaN in the output) without a message.  One-time check per cached weight (synchronises once)."""
        limit = 65504.0 / float(1 << shift)
        amax = float(w.abs().max()) if w.numel() else 0.0
        if not amax < limit:
            raise ValueError(f'{what}: max |w| = {amax:.4g} does not fit the fp16 operand planes (|w| < {limit:.4g} at the '
                             f'2^{shift} pre-scale; see include/unimatch_hip.h, "Operand range")')

    def _check_rows(self, name, t, k):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous() and t.shape[1] == k):
            raise ValueError(f'{name}: expected a contiguous CUDA float32 [M, {k}] tensor, got {tuple(t.shape)} {t.dtype}')

    def linear_planes(self, a, weights, a1=None, gelu=False, a_planes_k=None):
        """``(gelu)(A . cat(weights)^T)`` written as MFMA operand planes ``[NS][M][N]`` (flat uint8 tensor).

        A is ``a`` fp32 ``[M, K]``, or the K-concatenation of ``a`` and ``a1`` (``[M, K/2]`` each), or -- with
        ``a_planes_k`` -- ``a`` is already a plane tensor with K = a_planes_k columns."""
        wp, n, k = self.weight_planes(weights)
        if a_planes_k is not None:
            m = a.numel() // (2 * self.nplanes * a_planes_k)
            args = (None, None, _ptr(a))
        else:
            m = a.shape[0]
            self._check_rows('a', a, k // 2 if a1 is not None else k)
            if a1 is not None:
                self._check_rows('a1', a1, k // 2)
            args = (_ptr(a), _ptr(a1) if a1 is not None else None, None)
        out = torch.empty(self.lib.um_planes_bytes(m, n, self.mode), dtype=torch.uint8, device=a.device)
        meta = {'flops': 2.0 * m * n * k}
        code = self._launch('linear', lambda: self.lib.um_linear_fwd(
            args[0], args[1], args[2], _ptr(wp), m, n, k, self.WSHIFT, 2 if gelu else 0, _ptr(out),
            None, None, None, 0.0, self.mode, _stream()), meta)
        _abi.check(code, 'um_linear_fwd')
        return out, m, n

    def linear_ln(self, a, weights, norm, residual=None, a_planes_k=None):
        """``LayerNorm(A . W^T) (+ residual)`` -> fp32 ``[M, 128]``; ``norm`` is an ``nn.LayerNorm`` over 128."""
        wp, n, k = self.weight_planes(weights)
        if a_planes_k is not None:
            m = a.numel() // (2 * self.nplanes * a_planes_k)
            args = (None, None, _ptr(a))
        else:
            m = a.shape[0]
            self._check_rows('a', a, k)
            args = (_ptr(a), None, None)
        if residual is not None:
            self._check_rows('residual', residual, n)
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
        code = self._launch('linear', lambda: self.lib.um_linear_fwd(
            args[0], args[1], args[2], _ptr(wp), m, n, k, self.WSHIFT, 1, _ptr(out),
            _ptr(norm.weight), _ptr(norm.bias), _ptr(residual) if residual is not None else None,
            float(norm.eps), self.mode, _stream()), {'flops': 2.0 * m * n * k})
        _abi.check(code, 'um_linear_fwd')
        return out

    def ffn_ln(self, x, y, w1, w2, norm):
        """``x + LayerNorm(W2 . gelu(W1 . [x | y]))`` in one kernel (``um_ffn_fwd``); x, y fp32 ``[M, 128]``,
        ``w1`` ``[hidden, 256]``, ``w2`` ``[128, hidden]``, ``norm`` an ``nn.LayerNorm`` over 128."""
        w1p, hid, k1 = self.weight_planes((w1,))
        w2p, n2, k2 = self.weight_planes((w2,))
        if (k1, n2, k2) != (256, 128, hid):
            raise ValueError(f'ffn_ln: expected W1 [hidden, 256] and W2 [128, hidden], got {tuple(w1.shape)} {tuple(w2.shape)}')
        self._check_rows('x', x, 128)
        self._check_rows('y', y, 128)
        m = x.shape[0]
        if y.shape[0] != m:
            raise ValueError('ffn_ln: x and y must have the same number of rows')
        out = torch.empty((m, 128), dtype=torch.float32, device=x.device)
        ws = self._split_workspace('_ffn_ws', self.lib.um_ffn_split_workspace_bytes(m, hid), x.device)
        code = self._launch('ffn', lambda: self.lib.um_ffn_ws_fwd(
            _ptr(x), _ptr(y), _ptr(w1p), _ptr(w2p), m, hid, self.WSHIFT, _ptr(norm.weight), _ptr(norm.bias),
            float(norm.eps), _ptr(out), self.mode, _ptr(ws) if ws is not None else None, ws.numel() if ws is not None else 0,
            _stream()), {'flops': 2.0 * m * hid * (256 + 128)})
        _abi.check(code, 'um_ffn_ws_fwd')
        return out

    def ffn_ln_kv(self, x, y, w1, w2, norm, kv_weights):
        """:meth:`ffn_ln` plus the NEXT block's key / value projections of its result (:meth:`kv4_planes` of the returned tokens with
        ``kv_weights`` = next block's (k_self, v_self, k_cross, v_cross)) from the same launch (``um_ffn_kv_fwd``; small launches
        run the two kernels back to back inside the call) -> ``(out fp32 [M, 128], blocked k | v planes)``."""
        w1p, hid, k1 = self.weight_planes((w1,))
        w2p, n2, k2 = self.weight_planes((w2,))
        if (k1, n2, k2) != (256, 128, hid):
            raise ValueError(f'ffn_ln_kv: expected W1 [hidden, 256] and W2 [128, hidden], got {tuple(w1.shape)} {tuple(w2.shape)}')
        self._check_rows('x', x, 128)
        self._check_rows('y', y, 128)
        m = x.shape[0]
        if y.shape[0] != m:
            raise ValueError('ffn_ln_kv: x and y must have the same number of rows')
        wc = self.kv4_weight_planes(kv_weights)
        out = torch.empty((m, 128), dtype=torch.float32, device=x.device)
        kv = torch.empty(self.lib.um_planes_bytes(4 * m, 128, self.mode), dtype=torch.uint8, device=x.device)
        ws = self._split_workspace('_ffn_ws', self.lib.um_ffn_split_workspace_bytes(m, hid), x.device)
        code = self._launch('ffn', lambda: self.lib.um_ffn_kv_fwd(
            _ptr(x), _ptr(y), _ptr(w1p), _ptr(w2p), m, hid, self.WSHIFT, _ptr(norm.weight), _ptr(norm.bias),
            float(norm.eps), _ptr(out), _ptr(wc), _ptr(kv), self.mode, _ptr(ws) if ws is not None else None,
            ws.numel() if ws is not None else 0, _stream()), {'flops': 2.0 * m * hid * (256 + 128) + 2.0 * m * 512 * 128})
        _abi.check(code, 'um_ffn_kv_fwd')
        return out, kv

    def window_attention_planes(self, q, k, v, streams, h, w, win_h, win_w, shift_h=0, shift_w=0, kv_rotate=0):
        """Attention on operand planes.  q, k, v: ``(plane_tensor, rows, cols, col_offset)`` -- a 128-column slice
        starting at ``col_offset`` of a ``[NS][rows][cols]`` plane tensor (k and v must share their tensor's shape)."""
        (qt, qrows, qcols, qoff), (kt, krows, kcols, koff), (vt, vrows, vcols, voff) = q, k, v
        if (krows, kcols) != (vrows, vcols) or qrows != streams * h * w or krows != qrows:
            raise ValueError('inconsistent plane shapes')
        out = torch.empty((streams, h * w, 128), dtype=torch.float32, device=qt.device)
        n = win_h * win_w
        meta = {'flops': 4.0 * streams * h * w * n * 128}
        code = self._launch('window_attn', lambda: self.lib.um_window_attn_planes_fwd(
            _ptr(qt) + 2 * qoff, _ptr(kt) + 2 * koff, _ptr(vt) + 2 * voff, _ptr(out), streams, h, w, 128,
            qcols, kcols, qrows * qcols, krows * kcols, win_h, win_w, shift_h, shift_w, kv_rotate, self.mode, _stream()), meta)
        _abi.check(code, 'um_window_attn_planes_fwd')
        return out

    def window_attention_merge(self, q, k, v, streams, h, w, win_h, win_w, shift_h, shift_w, kv_rotate, merge_weight, norm,
                               residual=None):
        """Attention on operand planes with ``LayerNorm(message . Wm^T) (+ residual)`` folded into the kernel's epilogue
        (``um_window_attn_merge_fwd``); arguments as :meth:`window_attention_planes`."""
        (qt, qrows, qcols, qoff), (kt, krows, kcols, koff), (vt, vrows, vcols, voff) = q, k, v
        if (krows, kcols) != (vrows, vcols) or qrows != streams * h * w or krows != qrows:
            raise ValueError('inconsistent plane shapes')
        wp, n, kk = self.weight_planes((merge_weight,))
        if (n, kk) != (128, 128):
            raise ValueError('window_attention_merge: the merge weight must be [128, 128]')
        if residual is not None:
            self._check_rows('residual', residual, 128)
        out = torch.empty((streams, h * w, 128), dtype=torch.float32, device=qt.device)
        meta = {'flops': 4.0 * streams * h * w * win_h * win_w * 128}
        code = self._launch('window_attn', lambda: self.lib.um_window_attn_merge_fwd(
            _ptr(qt) + 2 * qoff, _ptr(kt) + 2 * koff, _ptr(vt) + 2 * voff, _ptr(wp), _ptr(norm.weight), _ptr(norm.bias),
            _ptr(residual) if residual is not None else None, float(norm.eps), self.WSHIFT, _ptr(out), streams, h, w, 128,
            qcols, kcols, qrows * qcols, krows * kcols, win_h, win_w, shift_h, shift_w, kv_rotate, self.mode, _stream()), meta)
        _abi.check(code, 'um_window_attn_merge_fwd')
        return out

    def _ksplit_workspace(self, nbytes, device):
        return self._split_workspace('_ks_ws', nbytes, device)

    def _split_workspace(self, name, nbytes, device):
        """Scratch of a split small launch (attention: key split, FFN: hidden split -- partial results + arrival counters): zero
        at allocation, left zero by every launch; one buffer per kernel, device AND owner, grown on demand.  The owner is the
        current stream (two launches in flight on different streams must not share the counters) -- or, while a
        ``GraphedUniMatch`` warms up and captures, that graph's token (``workspace_owner``): the buffer is then allocated and
        zeroed by the eager warm-up OUTSIDE the capture, the capture only bakes its address in, every graph has its own buffer
        (two graphs replayed concurrently share no counter), and ``claim_workspaces`` hands it to the graph object -- an aborted
        capture drops it, so a buffer a capture may have left with stale counters is never reused."""
        if not nbytes:
            return None
        cache = self.__dict__.setdefault('_split_ws', {})
        owner = self._owner()
        key = (name, device, owner)
        buf = cache.get(key)
        # EXACT size (round 6): the arrival counters sit at the head of the buffer and their count is the launch's tile count -- a bigger
        # buffer left by another geometry has that geometry's partial results where this launch's counters must be zero (found with a
        # batch of 3 as parts of 2 + 1 pairs on one stream: non-finite results; the same happened to ANY sequence of two small-launch
        # geometries through one HipOps).  A geometry change on a stream costs one zero-fill.
        if buf is None or buf.numel() != nbytes:
            if torch.cuda.is_current_stream_capturing():
                # First request INSIDE a capture (a user's own torch.cuda.graph around the model, or a wrapper GraphedUniMatch could
                # not install its owner token through): a capture-PRIVATE buffer -- allocated from the graph's pool, zeroed by a memset
                # node that replays with the graph, referenced by this capture only and never cached, so no later launch can meet
                # counters an aborted capture left behind.  (Round 4 raised here, which turned such captures into a permanent eager
                # fallback.)
                return torch.zeros(nbytes, dtype=torch.uint8, device=device)
            buf = cache[key] = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        return buf

    workspace_owner = None     # set by GraphedUniMatch around warm-up + capture (see _split_workspace)
    workspace_lane = 0         # set by streams.PartRunner around each part's forward: the parts of a batch run on different streams

    def _owner(self):
        """Who owns the per-launch scratch requested now: the current stream, or -- under a graph's token -- (token, lane): a captured
        forward that runs its batch as concurrent parts (UniMatch.forward, streams.PartRunner) has one set of arrival counters /
        partial buffers / activation planes PER PART (round-5 ADVICE: keyed by the token alone, the two halves shared them on two streams)."""
        if self.workspace_owner is not None:
            return (self.workspace_owner, self.workspace_lane)
        return _stream()

    def claim_workspaces(self, owner):
        """Remove and return every split workspace allocated under ``owner`` (the graph keeps them alive; nobody else can get them)."""
        cache = self.__dict__.setdefault('_split_ws', {})
        mine = {k: cache.pop(k) for k in [k for k in cache if isinstance(k[2], tuple) and k[2][0] is owner]}
        return mine

    def window_attention_qproj_merge(self, x, q_weight, k, v, streams, h, w, win_h, win_w, shift_h, shift_w, kv_rotate,
                                     merge_weight, norm, residual=None):
        """:meth:`window_attention_merge` with ``q = x . Wq^T`` computed in the kernel's prologue
        (``um_window_attn_qproj_merge_fwd``): ``x`` fp32 ``[streams*h*w, 128]`` source tokens, ``q_weight`` ``[128, 128]``."""
        # k, v: (plane tensor, rows, row stride in elements, element offset of the first row[, plane stride in elements]) -- the
        # optional fifth entry describes slices of a bigger plane tensor (the blocked [NS][4][M][128] k | v planes of kv4_planes)
        (kt, krows, kcols, koff), (vt, vrows, vcols, voff) = k[:4], v[:4]
        kps = k[4] if len(k) > 4 else krows * kcols
        if (v[4] if len(v) > 4 else vrows * vcols) != kps:
            raise ValueError('k and v must share their plane stride')
        self._check_rows('x', x, 128)
        if (krows, kcols) != (vrows, vcols) or x.shape[0] != streams * h * w or krows != x.shape[0]:
            raise ValueError('inconsistent plane shapes')
        wqp, nq, kq = self.weight_planes((q_weight,))
        wp, n, kk = self.weight_planes((merge_weight,))
        if (n, kk, nq, kq) != (128, 128, 128, 128):
            raise ValueError('window_attention_qproj_merge: the query and merge weights must be [128, 128]')
        if residual is not None:
            self._check_rows('residual', residual, 128)
        out = torch.empty((streams, h * w, 128), dtype=torch.float32, device=x.device)
        meta = {'flops': 4.0 * streams * h * w * win_h * win_w * 128}
        ks = self._ksplit_workspace(self.lib.um_window_attn_ksplit_workspace_bytes(streams, h, w, win_h, win_w), x.device)
        code = self._launch('window_attn', lambda: self.lib.um_window_attn_qproj_merge_fwd(
            _ptr(x), _ptr(wqp), _ptr(kt) + 2 * koff, _ptr(vt) + 2 * voff, _ptr(wp), _ptr(norm.weight), _ptr(norm.bias),
            _ptr(residual) if residual is not None else None, float(norm.eps), self.WSHIFT, _ptr(out), streams, h, w, 128,
            kcols, kps, win_h, win_w, shift_h, shift_w, kv_rotate, self.mode,
            _ptr(ks) if ks is not None else None, ks.numel() if ks is not None else 0, _stream()), meta)
        _abi.check(code, 'um_window_attn_qproj_merge_fwd')
        return out

    # ------------------------------------------------------------------ k | v projections of a whole Transformer block
    def kv4_weight_planes(self, weights):
        """Planes of the packed ``[256, 256]`` weight of ``um_kv4_fwd`` / ``um_ffn_kv_fwd`` from the four ``[128, 128]`` projection
        weights (k_self, v_self, k_cross, v_cross): ``Wc[32c + r, :128] = W4[32c + r]``, ``Wc[32c + r, 128:] = W4[256 + 32c + (r ^ 16)]``
        (include/unimatch_hip.h, um_kv4_fwd); cached until a weight changes."""
        key, hit = self._cache_get('kv4', weights)
        if hit is not None:
            return hit
        if len(weights) != 4 or any(tuple(w.shape) != (128, 128) for w in weights):
            raise ValueError('kv4_weight_planes: expected four [128, 128] weights (k_self, v_self, k_cross, v_cross)')
        wc = pack_kv4_weights(weights)                                               # [256, 256]
        self._check_weight_range(wc, self.WSHIFT if self.mode == 0 else 0, 'Linear weight')
        planes = torch.empty(self.lib.um_planes_bytes(256, 256, self.mode), dtype=torch.uint8, device=wc.device)
        _abi.check(self.lib.um_weight_planes(_ptr(wc), _ptr(planes), 256, 256, self.WSHIFT, self.mode, _stream()), 'um_weight_planes')
        return self._cache_put(key, weights, planes)

    @staticmethod
    def kv4_slices(kv, m):
        """The four projections of blocked k | v planes ``[NS][4][m][128]`` as attention operands: ``((k_self, v_self), (k_cross, v_cross))``."""
        part = lambda j: (kv, m, 128, j * m * 128, 4 * m * 128)
        return (part(0), part(1)), (part(2), part(3))

    def kv4_planes(self, x, weights):
        """``um_kv4_fwd``: both layers' key / value projections of a Transformer block (four bias-free 128 x 128 Linears of the
        token stream ``x`` fp32 ``[M, 128]``, transformer.py:58-60) in one launch -> blocked operand planes ``[NS][4][M][128]``."""
        self._check_rows('x', x, 128)
        wc = self.kv4_weight_planes(weights)
        m = x.shape[0]
        out = torch.empty(self.lib.um_planes_bytes(4 * m, 128, self.mode), dtype=torch.uint8, device=x.device)
        code = self._launch('linear', lambda: self.lib.um_kv4_fwd(_ptr(x), _ptr(wc), m, self.WSHIFT, _ptr(out), self.mode, _stream()),
                            {'flops': 2.0 * m * 512 * 128})
        _abi.check(code, 'um_kv4_fwd')
        return out

    # ------------------------------------------------------------------ convex upsampling (SURVEY 8(f) "next" row)
    def convex_upsample(self, flow, mask, factor, is_depth=False, mask_nhwc=False):
        """RAFT convex upsampling of ``flow [B,V,h,w]`` with ``mask [B,9*factor^2,h,w]`` (or NHWC ``[B*h*w, 9*factor^2]``
        with ``mask_nhwc``) -> ``[B,V,factor*h,factor*w]``."""
        b, v, h, w = flow.shape
        want = (b * h * w, 9 * factor * factor) if mask_nhwc else (b, 9 * factor * factor, h, w)
        if not (flow.is_cuda and flow.dtype == torch.float32 and mask.dtype == torch.float32 and tuple(mask.shape) == want):
            raise ValueError(f'convex_upsample: bad shapes flow {tuple(flow.shape)} mask {tuple(mask.shape)}')
        flow, mask = flow.contiguous(), mask.contiguous()
        up = torch.empty((b, v, factor * h, factor * w), dtype=torch.float32, device=flow.device)
        code = self._launch('convex_upsample', lambda: self.lib.um_convex_upsample(
            _ptr(flow), _ptr(mask), _ptr(up), b, v, h, w, factor, int(bool(is_depth)), int(bool(mask_nhwc)), _stream()))
        _abi.check(code, 'um_convex_upsample')
        return up

    def flow_warp(self, tokens, flow, h, w):
        """``um_flow_warp``: bilinear warp of token-major features ``[b, h*w, c]`` by ``flow [b,2,h,w]`` -> tokens."""
        b, l, c = tokens.shape
        if not (tokens.is_cuda and tokens.dtype == torch.float32 and tokens.is_contiguous() and l == h * w
                and tuple(flow.shape) == (b, 2, h, w) and flow.dtype == torch.float32):
            raise ValueError(f'flow_warp: bad shapes tokens {tuple(tokens.shape)} flow {tuple(flow.shape)}')
        flow = flow.contiguous()
        out = torch.empty_like(tokens)
        code = self._launch('convex_upsample', lambda: self.lib.um_flow_warp(
            _ptr(tokens), _ptr(flow), _ptr(out), b, h, w, c, _stream()))
        _abi.check(code, 'um_flow_warp')
        return out

    def flow_upsample2x(self, flow, mult=2.0):
        """``mult * F.interpolate(flow, scale_factor=2, mode='bilinear', align_corners=True)`` (``um_flow_upsample2x``)."""
        if not (flow.is_cuda and flow.dtype == torch.float32 and flow.dim() == 4):
            raise ValueError('flow_upsample2x: expected a CUDA float32 [B, V, h, w] tensor')
        flow = flow.contiguous()
        b, v, h, w = flow.shape
        out = torch.empty((b, v, 2 * h, 2 * w), dtype=torch.float32, device=flow.device)
        _abi.check(self.lib.um_flow_upsample2x(_ptr(flow), _ptr(out), b, v, h, w, float(mult), _stream()), 'um_flow_upsample2x')
        return out

    def depth_cam(self, intrinsics, pose, stride_div, bidir=False):
        """``[B or 2B, 30]`` = K^-1 | R | t | K of the depth kernels from the caller's intrinsics ``[B,3,3]`` (rows 0-1 divided by
        ``stride_div``) and pose ``[B,4,4]`` (``um_depth_cam_pack``; with ``bidir`` the second half carries the inverse pose)."""
        b = intrinsics.shape[0]
        if not (tuple(intrinsics.shape) == (b, 3, 3) and tuple(pose.shape) == (b, 4, 4) and intrinsics.is_cuda and pose.is_cuda):
            raise ValueError('depth_cam: expected CUDA intrinsics [B,3,3] and pose [B,4,4]')
        k, p = intrinsics.float().contiguous(), pose.float().contiguous()
        cam = torch.empty((2 * b if bidir else b, 30), dtype=torch.float32, device=k.device)
        _abi.check(self.lib.um_depth_cam_pack(_ptr(k), _ptr(p), _ptr(cam), b, float(stride_div), int(bool(bidir)), _stream()),
                   'um_depth_cam_pack')
        return cam

    def rigid_flow(self, inv_depth, cam):
        """Flow induced by inverse depth ``[B,1,h,w]`` and ``cam [B,30]`` (``um_rigid_flow``, geometry.py:99-195)."""
        b, _, h, w = inv_depth.shape
        if not (inv_depth.is_cuda and inv_depth.dtype == torch.float32 and tuple(cam.shape) == (b, 30)):
            raise ValueError('rigid_flow: expected CUDA float32 inv_depth [B,1,h,w] and cam [B,30]')
        inv_depth = inv_depth.contiguous()
        out = torch.empty((b, 2, h, w), dtype=torch.float32, device=inv_depth.device)
        _abi.check(self.lib.um_rigid_flow(_ptr(inv_depth), _ptr(cam), _ptr(out), b, h, w, _stream()), 'um_rigid_flow')
        return out

    # ------------------------------------------------------------------ encoder helper (outside the hot path)
    def instance_norm(self, x, relu=True, shortcut=None, eps=1e-5):
        """Fused InstanceNorm2d(affine=False) [+ ReLU] [+ shortcut, ReLU] on a contiguous NCHW fp32 map."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
            raise ValueError('instance_norm: expected a contiguous CUDA float32 NCHW tensor')
        if shortcut is not None and not (shortcut.shape == x.shape and shortcut.is_contiguous()
                                         and shortcut.dtype == torch.float32):
            raise ValueError('instance_norm: shortcut must match x')
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        code = self._launch('instance_norm', lambda: self.lib.um_instance_norm_fwd(
            _ptr(x), _ptr(shortcut) if shortcut is not None else None, _ptr(y), n * c, h * w, float(eps),
            int(bool(relu)), _stream()))
        _abi.check(code, 'um_instance_norm_fwd')
        return y

    # ------------------------------------------------------------------ NHWC convolutions on the matrix cores
    # (also what the encoder uses).  An "NHWC activation" here is a small record:
    #   planes: operand planes [NS][rows + 1][C] (uint8 tensor; last row zero) or None
    #   f32   : fp32 [rows, C] or None;   b, h, w, c: geometry, rows = b * h * w
    def conv_weight_planes(self, weight):
        """Planes of an ``nn.Conv2d`` weight ``[cout, cin, kh, kw]`` permuted to ``[cout, kh*kw*cin]``; cached."""
        key, hit = self._cache_get('conv', (weight,))
        if hit is not None:
            return hit
        cout, cin, kh, kw = weight.shape
        w2 = weight.detach().float().permute(0, 2, 3, 1).reshape(cout, kh * kw * cin).contiguous()
        planes = torch.empty(self.lib.um_planes_bytes(cout, kh * kw * cin, self.CONV_MODE), dtype=torch.uint8, device=w2.device)
        self._check_weight_range(w2, self.WSHIFT, 'convolution weight')
        _abi.check(self.lib.um_weight_planes(_ptr(w2), _ptr(planes), cout, kh * kw * cin, self.WSHIFT, self.CONV_MODE, _stream()),
                   'um_weight_planes')
        return self._cache_put(key, (weight,), (planes, cout, cin, kh, kw))

    def conv2d_nhwc(self, act, weight, bias=None, stride=1, padding=(1, 1), relu=False, stats=False):
        """``act``: ``(planes, b, h, w, cin)``; returns fp32 ``[b*ho*wo, cout]`` and ``(ho, wo)``.  With ``stats`` the epilogue also emits the per-tile InstanceNorm statistics: ``self.last_conv_stats``."""
        planes, b, h, w, cin = act
        wp, cout, wcin, kh, kw = self.conv_weight_planes(weight)
        if wcin != cin:
            raise ValueError(f'conv2d_nhwc: weight expects {wcin} input channels, activation has {cin}')
        ph, pw = (padding, padding) if isinstance(padding, int) else padding
        ho, wo = (h + 2 * ph - kh) // stride + 1, (w + 2 * pw - kw) // stride + 1
        out = torch.empty((b * ho * wo, cout), dtype=torch.float32, device=planes.device)
        self.last_conv_stats = None
        if stats:                                    # (per-part statistics, parts per image): hand both to nhwc_norm(conv_stats=)
            parts = self.lib.um_conv_stats_parts(h, w, cout, kh, kw, stride, ph, pw)
            self.last_conv_stats = (torch.empty(self.lib.um_conv_stats_bytes(b, parts, cout) // 4, dtype=torch.float32,
                                                device=planes.device), parts)
        st = self.last_conv_stats[0] if self.last_conv_stats is not None else None
        meta = {'flops': 2.0 * b * ho * wo * cout * kh * kw * cin}
        code = self._launch('conv', lambda: self.lib.um_conv2d_fwd(
            _ptr(planes), _ptr(wp), _ptr(bias) if bias is not None else None, _ptr(out), _ptr(st) if st is not None else None,
            b, h, w, cin, cout, kh, kw,
            stride, ph, pw, int(bool(relu)), self.WSHIFT, self.CONV_MODE, _stream()), meta)
        _abi.check(code, 'um_conv2d_fwd')
        return out, ho, wo

    def nhwc_planes_from(self, pieces, pad_to=32):
        """Operand planes of the channel concatenation of fp32 NHWC pieces ``[rows, c_i]`` (zero-padded to a multiple of
        ``pad_to`` channels): returns ``(planes, channels)``."""
        rows = pieces[0].shape[0]
        c = sum(p.shape[1] for p in pieces)
        cp = (c + pad_to - 1) // pad_to * pad_to
        if cp != c:
            pieces = list(pieces) + [torch.zeros((rows, cp - c), dtype=torch.float32, device=pieces[0].device)]
        x = pieces[0].contiguous() if len(pieces) == 1 else torch.cat(list(pieces), 1)
        planes, _ = self.nhwc_norm(x, 1, rows, normalize=False, relu=False, want_planes=True)
        return planes, cp

    def conv_weight_padded(self, weight, cin_to):
        """``weight [cout, cin, kh, kw]`` with zero input channels appended up to ``cin_to`` (cached by the caller's
        parameter identity through conv_weight_planes)."""
        key, hit = self._cache_get(('padw', cin_to), (weight,))
        if hit is None:
            cout, cin, kh, kw = weight.shape
            wpad = torch.zeros((cout, cin_to, kh, kw), dtype=torch.float32, device=weight.device)
            wpad[:, :cin] = weight.detach().float()
            hit = self._cache_put(key, (weight,), wpad)
        return hit

    # ---- building blocks of the channels-last refinement block (unimatch_amd/refine_nhwc.py) ---------------------------------
    def planes_buffer(self, rows, ld):
        """Zeroed operand-plane buffer ``[2][rows + 1][ld]`` (fp16 hi | lo); row ``rows`` is the zero padding row."""
        return torch.zeros(2 * (rows + 1) * ld * 2, dtype=torch.uint8, device='cuda')

    def cached_planes_buffer(self, tag, rows, ld, device):
        """A ``planes_buffer(rows, ld)`` that is allocated and zeroed ONCE per (tag, geometry, device, owner) and handed out again on
        later forwards (the refinement block's six activation buffers are 590 MB of zero-fill per forward at config 4 otherwise).
        Only the padding row has to be zero and no kernel ever writes it; every other row is written before it is read.  Ownership
        as for the split workspaces (:meth:`_split_workspace`): per stream, or per captured graph."""
        name = ('planes', tag, rows, ld)
        cache = self.__dict__.setdefault('_split_ws', {})
        owner = self._owner()
        device = torch.device(device)
        for k in [k for k in cache if isinstance(k[0], tuple) and k[0][:2] == name[:2] and k[0] != name and k[2] == owner
                  and torch.device(k[1]) == device]:
            del cache[k]                       # one geometry per (owner, device) stays resident (a new input size replaces the old set)
        if (name, device, owner) not in cache and torch.cuda.is_current_stream_capturing():
            return self.planes_buffer(rows, ld)  # captured without a warm-up under an owner: the graph's own zero-filled buffer
        return self._split_workspace(name, 2 * (rows + 1) * ld * 2, device)

    def release_cached_planes(self):
        """Drop every cached activation-plane buffer of this instance (the refinement block keeps ~590 MB per stream at config 4
        alive between forwards; captured graphs keep their own sets until the graph object dies).  Safe at any time between
        forwards: the next forward allocates and zeroes a fresh set."""
        cache = self.__dict__.setdefault('_split_ws', {})
        for k in [k for k in cache if isinstance(k[0], tuple) and k[0][0] == 'planes']:
            del cache[k]

    def conv_weight_planes_from(self, weight):
        """Uncached: planes of ``weight [cout, cin, kh, kw]`` permuted to ``[cout, kh*kw*cin]`` -> ``(planes, cout, cin, kh, kw)``."""
        cout, cin, kh, kw = weight.shape
        w2 = weight.detach().float().permute(0, 2, 3, 1).reshape(cout, kh * kw * cin).contiguous()
        planes = torch.empty(self.lib.um_planes_bytes(cout, kh * kw * cin, 0), dtype=torch.uint8, device=w2.device)
        self._check_weight_range(w2, self.WSHIFT, 'convolution weight')
        _abi.check(self.lib.um_weight_planes(_ptr(w2), _ptr(planes), cout, kh * kw * cin, self.WSHIFT, 0, _stream()),
                   'um_weight_planes')
        return planes, cout, cin, kh, kw

    def conv_ex(self, src, geom, wb, ksize, stride, pad, act, out=None, outp=None):
        """``um_conv2d_ex``.  ``src = (planes_buffer, ld, coff, cin)``, ``geom = (b, h, w)``, ``wb = ((w_planes, cout, cin, kh,
        kw), bias)``, ``out = (fp32 tensor, ld, coff)``, ``outp = (planes_buffer, ld, coff)``; act 0/1/2/3 = none/ReLU/sigmoid/tanh."""
        buf, a_ld, a_coff, cin = src
        b, h, w = geom
        (wp, cout, wcin, kh, kw), bias = wb
        if wcin != cin or (kh, kw) != tuple(ksize):
            raise ValueError(f'conv_ex: weight is {cout}x{wcin}x{kh}x{kw}, call wants cin={cin} k={ksize}')
        a_rows = buf.numel() // (4 * a_ld)
        ho, wo = (h + 2 * pad[0] - kh) // stride + 1, (w + 2 * pad[1] - kw) // stride + 1
        o_t, o_ld, o_coff = out if out is not None else (None, 0, 0)
        p_t, p_ld, p_coff = outp if outp is not None else (None, 0, 0)
        p_rows = p_t.numel() // (4 * p_ld) if p_t is not None else 0
        code = self._launch('conv', lambda: self.lib.um_conv2d_ex(
            _ptr(buf), a_ld, a_coff, a_rows, _ptr(wp), _ptr(bias) if bias is not None else None,
            _ptr(o_t) if o_t is not None else None, o_ld, o_coff, _ptr(p_t) if p_t is not None else None, p_ld, p_coff, p_rows,
            None, b, h, w, cin, cout, kh, kw, stride, pad[0], pad[1], act, self.WSHIFT, 0, _stream()),
            {'flops': 2.0 * b * ho * wo * cout * kh * kw * cin})
        _abi.check(code, 'um_conv2d_ex')

    def conv_gru(self, gate, src, geom, wb, ksize, pad, hidden, outp, z=None, z_out=None, addend=None, hidden_out=None):
        """``um_conv2d_gru_fwd``: gate 1 = (z | r) convolution -> ``z_out`` fp32 and ``r * hidden`` planes; gate 2 = q
        convolution -> ``hidden`` updated in place and written as planes.  ``src`` / ``wb`` / ``outp`` as :meth:`conv_ex`.
        ``addend``: fp32 ``[rows, cout]`` added before the gate's activation (``um_conv2d_gru_add_fwd``: the iteration-invariant
        input channels' share of the convolution, computed once per scale).  ``hidden_out`` (gate 2 with an addend): the new state goes
        there and ``hidden`` is only read."""
        buf, a_ld, a_coff, cin = src
        b, h, w = geom
        (wp, cout, wcin, kh, kw), bias = wb
        c = hidden.shape[1]
        if wcin != cin or (kh, kw) != tuple(ksize) or cout != (2 * c if gate == 1 else c):
            raise ValueError('conv_gru: weight / hidden shapes do not match')
        p_t, p_ld, p_coff = outp
        zt = z if gate == 2 else None
        if hidden_out is not None:
            if gate != 2 or addend is None or hidden_out.shape != hidden.shape or hidden_out.dtype != torch.float32 or not hidden_out.is_contiguous():
                raise ValueError('conv_gru: hidden_out goes with gate 2 and an addend, contiguous fp32 of the hidden state\'s shape')
            z_out = hidden_out
        if addend is not None:
            if not (addend.dtype == torch.float32 and addend.is_contiguous() and addend.dim() == 2 and addend.shape[1] == cout
                    and addend.shape[0] == geom[0] * geom[1] * geom[2]):
                raise ValueError(f'conv_gru: addend must be contiguous fp32 [b * h * w = {geom[0] * geom[1] * geom[2]}, cout = {cout}], '
                                 f'got {tuple(addend.shape)}')
            code = self._launch('conv', lambda: self.lib.um_conv2d_gru_add_fwd(
                gate, _ptr(buf), a_ld, a_coff, buf.numel() // (4 * a_ld), _ptr(wp), _ptr(bias) if bias is not None else None,
                _ptr(addend), addend.shape[1], _ptr(hidden), _ptr(zt) if zt is not None else None,
                zt.shape[1] if zt is not None else 0, _ptr(z_out) if z_out is not None else None,
                z_out.shape[1] if z_out is not None else 0, _ptr(p_t), p_ld, p_coff, p_t.numel() // (4 * p_ld), b, h, w, cin, c,
                kh, kw, pad[0], pad[1], self.WSHIFT, 0, _stream()), {'flops': 2.0 * b * h * w * cout * kh * kw * cin})
            _abi.check(code, 'um_conv2d_gru_add_fwd')
            return
        code = self._launch('conv', lambda: self.lib.um_conv2d_gru_fwd(
            gate, _ptr(buf), a_ld, a_coff, buf.numel() // (4 * a_ld), _ptr(wp), _ptr(bias) if bias is not None else None,
            _ptr(hidden), _ptr(zt) if zt is not None else None, zt.shape[1] if zt is not None else 0,
            _ptr(z_out) if z_out is not None else None, z_out.shape[1] if z_out is not None else 0, _ptr(p_t), p_ld, p_coff,
            p_t.numel() // (4 * p_ld), b, h, w, cin, c, kh, kw, pad[0], pad[1], self.WSHIFT, 0, _stream()),
            {'flops': 2.0 * b * h * w * cout * kh * kw * cin})
        _abi.check(code, 'um_conv2d_gru_fwd')

    def conv7(self, image, weight, bias, stride, act, out=None, outp=None):
        """7x7 / pad 3 convolution of an fp32 NCHW image with few channels (``um_conv7_fwd``), outputs as :meth:`conv_ex`."""
        b, c, h, w = image.shape
        cout = weight.shape[0]
        cpp = 4 if stride == 2 else 8
        key, hit = self._cache_get(('conv7', stride), (weight,))
        if hit is None:
            wr = torch.zeros((cout, 7, 8, cpp), dtype=torch.float32, device=weight.device)
            wr[:, :, :7, :c] = weight.detach().float().permute(0, 2, 3, 1)
            wr = wr.reshape(cout, 56 * cpp).contiguous()
            wp = torch.empty(self.lib.um_planes_bytes(cout, 56 * cpp, 0), dtype=torch.uint8, device=weight.device)
            self._check_weight_range(wr, self.WSHIFT, 'convolution weight')
            _abi.check(self.lib.um_weight_planes(_ptr(wr), _ptr(wp), cout, 56 * cpp, self.WSHIFT, 0, _stream()), 'um_weight_planes')
            hit = self._cache_put(key, (weight,), wp)
        scratch = torch.empty(self.lib.um_conv7_planes_bytes(b, h, w, stride), dtype=torch.uint8, device=image.device)
        o_t, o_ld, o_coff = out if out is not None else (None, 0, 0)
        p_t, p_ld, p_coff = outp if outp is not None else (None, 0, 0)
        p_rows = p_t.numel() // (4 * p_ld) if p_t is not None else 0
        image = image.contiguous()
        code = self._launch('conv', lambda: self.lib.um_conv7_fwd(
            _ptr(image), c, 0, None, None, _ptr(scratch), _ptr(hit), _ptr(bias) if bias is not None else None,
            _ptr(o_t) if o_t is not None else None, o_ld, o_coff, _ptr(p_t) if p_t is not None else None, p_ld, p_coff, p_rows,
            None, b, h, w, cout, stride, act, self.WSHIFT, _stream()))
        _abi.check(code, 'um_conv7_fwd')

    def nhwc_gate(self, mode, src, dest, ld, coff, rows, channels, zr=None, hbuf=None):
        """``um_nhwc_gate``: column scatter (0), ``r * h`` (1), GRU state update (2) into ``dest`` planes ``[2][.][ld]``."""
        p_rows = dest.numel() // (4 * ld) if dest is not None else 0
        src_ld = src.shape[1] if src is not None else 0
        code = self._launch('instance_norm', lambda: self.lib.um_nhwc_gate(
            mode, _ptr(src) if src is not None else None, src_ld, _ptr(zr) if zr is not None else None,
            _ptr(hbuf) if hbuf is not None else None, _ptr(dest) if dest is not None else None, ld, coff, p_rows, rows, channels,
            _stream()))
        _abi.check(code, 'um_nhwc_gate')

    def local_corr_with_flow_planes(self, f0, f1, flow, h, w, radius, dest, ld):
        """K4 written channels-last as operand planes (``um_local_corr_with_flow_planes``)."""
        b, l, c = f0.shape
        _check_tokens('f0', f0, tokens=h * w)
        _check_tokens('f1', f1, b, l)
        flow = flow.contiguous()
        feat = self._k4_feat_planes(f0, f1, h, w, radius)
        if feat is not None:
            code = self._launch('local_corr_with_flow', lambda: self.lib.um_local_corr_with_flow_feat(
                _ptr(f0), _ptr(f1), _ptr(feat), _ptr(flow), None, _ptr(dest), ld, dest.numel() // (4 * ld), b, h, w, c, radius,
                self.k4_flags, None, _stream()))
            _abi.check(code, 'um_local_corr_with_flow_feat')
            return
        code = self._launch('local_corr_with_flow', lambda: self.lib.um_local_corr_with_flow_planes(
            _ptr(f0), _ptr(f1), _ptr(flow), _ptr(dest), ld, dest.numel() // (4 * ld), b, h, w, c, radius, _stream()))
        _abi.check(code, 'um_local_corr_with_flow_planes')

    # Cost-volume dispatch is a PURE FUNCTION of the call's arguments: radius 4 on a map of whole 8 x 4 pixel tiles goes to the
    # matrix-core kernel (k4m_kernel), whose own per-tile test -- does the tile's flow fit one 32 x 24 window? -- picks the
    # product path or the gather path from the flow values alone; every other geometry goes to the VALU kernel.  Two forwards
    # on the same inputs therefore run the same instructions (bitwise-equal outputs, HIP-graph capture freezes nothing).
    # Round 2 routed launches by a tile-coherence counter read back asynchronously; that made the choice timing dependent
    # for ~3 % on incoherent flow (k4m's gather path 0.661 ms against 0.643 ms at 4 x 128 x 192) and is gone.
    k4_mfma = True             # False: VALU kernels everywhere (tests compare the two; tools A/B them)
    k4_flags = 0               # bit 0: force k4m_kernel's per-pixel path for every tile (diagnostics)

    def _k4_feat_planes(self, f0, f1, h, w, radius):
        """fp16 hi | lo operand planes of (f0, f1) for um_local_corr_with_flow_feat, or None where that kernel does not apply.
        The refinement loop passes the same two tensors in every iteration: one entry, keyed by identity and version."""
        if not self.k4_mfma or not self.lib.um_local_corr_with_flow_feat_supported(h, w, f0.shape[2], radius):
            return None
        key = (id(f0), f0._version, f0.data_ptr(), id(f1), f1._version, f1.data_ptr(), tuple(f0.shape))
        hit = getattr(self, '_k4_feat', None)
        if hit is not None and hit[0] == key and hit[1]() is f0 and hit[2]() is f1:
            return hit[3]
        b, l, c = f0.shape
        feat = torch.empty(self.lib.um_local_corr_feat_planes_bytes(b, h, w, c), dtype=torch.uint8, device=f0.device)
        _abi.check(self.lib.um_local_corr_feat_planes(_ptr(f0), _ptr(f1), _ptr(feat), b, h, w, c, _stream()),
                   'um_local_corr_feat_planes')
        self._k4_feat = (key, weakref.ref(f0), weakref.ref(f1), feat)
        return feat

    def stem_conv(self, image, weight, norm_mean_std=None, stats=True):
        """The encoder's 7x7/2 stem on ``um_stem_conv_fwd``: fp32 NCHW image ``[b,3,h,w]`` -> fp32 NHWC ``[b*ho*wo, cout]``.
        ``norm_mean_std``: ``((m0,m1,m2), (s0,s1,s2))`` applies the reference's ``(x / 255 - mean) / std`` while packing."""
        if not (image.is_cuda and image.dtype == torch.float32 and image.dim() == 4 and image.shape[1] == 3
                and image.is_contiguous()):
            raise ValueError('stem_conv: expected a contiguous CUDA float32 [b, 3, h, w] image')
        if tuple(weight.shape[1:]) != (3, 7, 7):
            raise ValueError('stem_conv: expected a [cout, 3, 7, 7] weight')
        b, _, h, w = image.shape
        cout = weight.shape[0]
        key, hit = self._cache_get('stem', (weight,))
        if hit is None:
            wr = torch.zeros((cout, 7, 8, 4), dtype=torch.float32, device=weight.device)
            wr[:, :, :7, :3] = weight.detach().float().permute(0, 2, 3, 1)
            wr = wr.reshape(cout, 224).contiguous()
            wp = torch.empty(self.lib.um_planes_bytes(cout, 224, 0), dtype=torch.uint8, device=weight.device)
            self._check_weight_range(wr, self.WSHIFT, 'convolution weight')
            _abi.check(self.lib.um_weight_planes(_ptr(wr), _ptr(wp), cout, 224, self.WSHIFT, 0, _stream()), 'um_weight_planes')
            hit = self._cache_put(key, (weight,), wp)
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        scratch = torch.empty(self.lib.um_stem_planes_bytes(b, h, w), dtype=torch.uint8, device=image.device)
        out = torch.empty((b * ho * wo, cout), dtype=torch.float32, device=image.device)
        self.last_conv_stats = None
        if stats:
            parts = self.lib.um_conv_stats_parts(h, w, cout, 7, 7, 2, 3, 3)
            self.last_conv_stats = (torch.empty(self.lib.um_conv_stats_bytes(b, parts, cout) // 4, dtype=torch.float32,
                                                device=image.device), parts)
        st = self.last_conv_stats[0] if self.last_conv_stats is not None else None
        if norm_mean_std is not None:
            mean = (ctypes.c_float * 3)(*[float(v) for v in norm_mean_std[0]])
            std = (ctypes.c_float * 3)(*[float(v) for v in norm_mean_std[1]])
        else:
            mean = std = None
        code = self._launch('conv', lambda: self.lib.um_stem_conv_fwd(
            _ptr(image), int(norm_mean_std is not None), mean, std, _ptr(scratch), _ptr(hit), _ptr(out),
            _ptr(st) if st is not None else None, b, h, w, cout, self.WSHIFT, _stream()),
            {'flops': 2.0 * b * ho * wo * cout * 147})
        _abi.check(code, 'um_stem_conv_fwd')
        return out, ho, wo

    def nhwc_norm(self, x, b, pixels, normalize=True, relu=True, shortcut=None, want_planes=True, want_f32=False, eps=1e-5,
                  conv_stats=None, shortcut_planes=None):
        """InstanceNorm (+ ReLU, + shortcut + ReLU) of fp32 NHWC ``x [b*pixels, c]`` -> ``(planes | None, f32 | None)``.
        The shortcut is fp32 ``[b*pixels, c]`` or (``shortcut_planes``) operand planes of the same shape; ``conv_stats`` is the
        ``(statistics, parts per image)`` pair the producing convolution left in ``last_conv_stats``."""
        self._check_rows('x', x, x.shape[1])
        c = x.shape[1]
        if x.shape[0] != b * pixels:
            raise ValueError('nhwc_norm: x must have b * pixels rows')
        if shortcut is not None:
            self._check_rows('shortcut', shortcut, c)
        rows = b * pixels
        if shortcut_planes is not None and (shortcut is not None or shortcut_planes.numel() != self.lib.um_planes_bytes(rows + 1, c, self.CONV_MODE)):
            raise ValueError('nhwc_norm: shortcut_planes must be operand planes of [b * pixels + 1, c] (and exclude shortcut)')
        planes = (torch.empty(self.lib.um_planes_bytes(rows + 1, c, self.CONV_MODE), dtype=torch.uint8, device=x.device)
                  if want_planes else None)
        f32 = torch.empty_like(x) if want_f32 else None
        ws = self._ws(self.lib.um_nhwc_norm_workspace_bytes(b, pixels, c), x.device) if normalize else None
        code = self._launch('instance_norm', lambda: self.lib.um_nhwc_instance_norm(
            _ptr(x), _ptr(shortcut) if shortcut is not None else None,
            _ptr(shortcut_planes) if shortcut_planes is not None else None, _ptr(planes) if planes is not None else None,
            _ptr(f32) if f32 is not None else None, b, pixels, c, float(eps), int(bool(normalize)), int(bool(relu)),
            _ptr(conv_stats[0]) if conv_stats is not None else None, conv_stats[1] if conv_stats is not None else 0,
            _ptr(ws) if ws is not None else None, ws.numel() if ws is not None else 0, self.CONV_MODE, _stream()))
        _abi.check(code, 'um_nhwc_instance_norm')
        return planes, f32

    def nchw_to_nhwc(self, x, want_planes=True, want_f32=False):
        """fp32 NCHW map -> NHWC operand planes and / or fp32 ``[b*h*w, c]``."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
            raise ValueError('nchw_to_nhwc: expected a contiguous CUDA float32 NCHW tensor')
        b, c, h, w = x.shape
        rows = b * h * w
        planes = (torch.empty(self.lib.um_planes_bytes(rows + 1, c, self.CONV_MODE), dtype=torch.uint8, device=x.device)
                  if want_planes else None)
        f32 = torch.empty((rows, c), dtype=torch.float32, device=x.device) if want_f32 else None
        code = self._launch('instance_norm', lambda: self.lib.um_nchw_to_nhwc(
            _ptr(x), _ptr(planes) if planes is not None else None, _ptr(f32) if f32 is not None else None, b, c, h * w,
            self.CONV_MODE, _stream()))
        _abi.check(code, 'um_nchw_to_nhwc')
        return planes, f32

    # ------------------------------------------------------------------ global matching
    def global_corr_softmax_flow(self, f0, f1, h, w, bidir=False):
        b, l, c = f0.shape
        _check_tokens('f0', f0, tokens=h * w)
        _check_tokens('f1', f1, b, l)
        out = torch.empty((2 * b if bidir else b, 2, h, w), dtype=torch.float32, device=f0.device)
        ws = self._ws(self.lib.um_global_corr_workspace_bytes(b, l, c, self.mode), f0.device)
        meta = {'flops': (2.0 if bidir else 1.0) * b * (2.0 * l * l * c + 4.0 * l * l),
                'bytes': 2.0 * 4 * b * l * c + 8.0 * b * l}
        code = self._launch('global_corr_flow', lambda: self.lib.um_global_corr_softmax_flow(
            _ptr(f0), _ptr(f1), _ptr(out), b, h, w, c, int(bool(bidir)), self.mode,
            _ptr(ws), ws.numel(), _stream()), meta)
        _abi.check(code, 'um_global_corr_softmax_flow')
        return out

    def global_corr_softmax_stereo(self, f0, f1, h, w):
        b, l, c = f0.shape
        _check_tokens('f0', f0, tokens=h * w)
        _check_tokens('f1', f1, b, l)
        out = torch.empty((b, 1, h, w), dtype=torch.float32, device=f0.device)
        ws = self._ws(self.lib.um_global_corr_workspace_bytes(b, l, c, self.mode), f0.device)
        code = self._launch('global_corr_stereo', lambda: self.lib.um_global_corr_softmax_stereo(
            _ptr(f0), _ptr(f1), _ptr(out), b, h, w, c, self.mode, _ptr(ws), ws.numel(), _stream()))
        _abi.check(code, 'um_global_corr_softmax_stereo')
        return out

    def prop_global(self, q, k, value, h, w):
        b, l, c = q.shape
        _check_tokens('q', q, tokens=h * w)
        _check_tokens('k', k, b, l)
        _check_map('value', value, b, h, w)
        out = torch.empty_like(value)
        ws = self._ws(self.lib.um_global_corr_workspace_bytes(b, l, c, self.mode), q.device)
        vch = value.shape[1]
        meta = {'flops': b * (2.0 * l * l * c + 2.0 * l * l * vch), 'bytes': 2.0 * 4 * b * l * c + 8.0 * b * l * vch}
        code = self._launch('prop_global', lambda: self.lib.um_prop_global_attn(
            _ptr(q), _ptr(k), _ptr(value), _ptr(out), b, h, w, c, vch, self.mode,
            _ptr(ws), ws.numel(), _stream()), meta)
        _abi.check(code, 'um_prop_global_attn')
        return out

    def linear_bias(self, a, weight, bias, out_mul=1.0, bias_mul=1.0, planes=False, a_planes_k=None):
        """``nn.Linear`` with bias on ``um_linear_bias_fwd``: ``(A . W^T) * out_mul + bias * bias_mul`` as fp32 ``[M, N]`` or
        (``planes``) as MFMA operand planes.  ``a``: fp32 ``[M, K]``, or planes with ``a_planes_k`` columns."""
        wp, n, k = self.weight_planes((weight,))
        if a_planes_k is not None:
            m = a.numel() // (2 * self.nplanes * a_planes_k)
            src = (None, _ptr(a))
        else:
            m = a.shape[0]
            self._check_rows('a', a, k)
            src = (_ptr(a), None)
        bias = bias.detach()
        if not (bias.is_cuda and bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == n):
            raise ValueError(f'linear_bias: expected a contiguous CUDA float32 bias of {n} elements')
        if planes:
            out = torch.empty(self.lib.um_planes_bytes(m, n, self.mode), dtype=torch.uint8, device=a.device)
            dst = (_ptr(out), None)
        else:
            out = torch.empty((m, n), dtype=torch.float32, device=a.device)
            dst = (None, _ptr(out))
        code = self._launch('linear', lambda: self.lib.um_linear_bias_fwd(
            src[0], src[1], _ptr(wp), _ptr(bias), m, n, k, self.WSHIFT, float(out_mul), float(bias_mul), dst[0], dst[1],
            self.mode, _stream()), {'flops': 2.0 * m * n * k})
        _abi.check(code, 'um_linear_bias_fwd')
        return out

    def prop_global_projected(self, tokens, q_proj, k_proj, value, h, w):
        """The whole global propagation layer (attention.py:196-213): ``q = Wq x + bq``, ``k = Wk q + bk`` (of q -- the
        reference's quirk), ``softmax(q k^T / sqrt(C)) value``.  q and k go from the projection kernels to the attention
        kernel as operand planes; no library GEMM, no fp32 q / k."""
        b, l, c = tokens.shape
        _check_tokens('tokens', tokens, tokens=h * w)
        _check_map('value', value, b, h, w)
        ps = self.lib.um_global_corr_plane_scale(c)
        x = tokens.reshape(b * l, c)
        qp = self.linear_bias(x, q_proj.weight, q_proj.bias, out_mul=ps, bias_mul=ps, planes=True)
        kp = self.linear_bias(qp, k_proj.weight, k_proj.bias, out_mul=1.0, bias_mul=ps, planes=True, a_planes_k=c)
        out = torch.empty_like(value)
        ws = self._ws(self.lib.um_global_corr_workspace_bytes(b, l, c, self.mode), tokens.device)
        vch = value.shape[1]
        meta = {'flops': b * (2.0 * l * l * c + 2.0 * l * l * vch), 'bytes': 2.0 * 4 * b * l * c + 8.0 * b * l * vch}
        code = self._launch('prop_global', lambda: self.lib.um_prop_global_attn_planes(
            _ptr(qp), _ptr(kp), _ptr(value), _ptr(out), b, h, w, c, vch, self.mode, _ptr(ws), ws.numel(), _stream()), meta)
        _abi.check(code, 'um_prop_global_attn_planes')
        return out

    # ------------------------------------------------------------------ local kernels (fp32)
    def local_corr_softmax(self, f0, f1, h, w, radius, one_d=False):
        b, l, c = f0.shape
        _check_tokens('f0', f0, tokens=h * w)
        _check_tokens('f1', f1, b, l)
        out = torch.empty((b, 1 if one_d else 2, h, w), dtype=torch.float32, device=f0.device)
        if not one_d and self.k4_mfma and self.lib.um_local_corr_with_flow_feat_supported(h, w, c, radius):
            ws = self._ws(self.lib.um_local_corr_feat_planes_bytes(b, h, w, c), f0.device)
            code = self._launch('local_corr_softmax', lambda: self.lib.um_local_corr_softmax_mfma(
                _ptr(f0), _ptr(f1), _ptr(out), b, h, w, c, radius, _ptr(ws), ws.numel(), _stream()))
            _abi.check(code, 'um_local_corr_softmax_mfma')
            return out
        code = self._launch('local_corr_softmax', lambda: self.lib.um_local_corr_softmax(
            _ptr(f0), _ptr(f1), _ptr(out), b, h, w, c, radius, int(bool(one_d)), _stream()))
        _abi.check(code, 'um_local_corr_softmax')
        return out

    def local_corr_with_flow(self, f0, f1, flow, h, w, radius, dilation=1):
        """``local_correlation_with_flow`` (matching.py:86-123) -> ``[B, (2r+1)^2, h, w]``; ``dilation`` as in the reference
        (1 everywhere in its callers; other values take ``um_local_corr_with_flow_dilated``'s plain gather)."""
        b, l, c = f0.shape
        _check_tokens('f0', f0, tokens=h * w)
        _check_tokens('f1', f1, b, l)
        _check_map('flow', flow, b, h, w)
        if flow.shape[1] != 2:
            raise ValueError('flow must have 2 channels')
        k = 2 * radius + 1
        out = torch.empty((b, k * k, h, w), dtype=torch.float32, device=f0.device)
        meta = {'flops': 2.0 * b * l * (k + 1) ** 2 * c, 'bytes': 2.0 * 4 * b * l * c + 8.0 * b * l + 4.0 * k * k * b * l}
        if dilation != 1:
            code = self._launch('local_corr_with_flow', lambda: self.lib.um_local_corr_with_flow_dilated(
                _ptr(f0), _ptr(f1), _ptr(flow.contiguous()), _ptr(out), b, h, w, c, radius, int(dilation), _stream()), meta)
            _abi.check(code, 'um_local_corr_with_flow_dilated')
            return out
        feat = self._k4_feat_planes(f0, f1, h, w, radius)
        if feat is not None:
            code = self._launch('local_corr_with_flow', lambda: self.lib.um_local_corr_with_flow_feat(
                _ptr(f0), _ptr(f1), _ptr(feat), _ptr(flow), _ptr(out), None, 0, 0, b, h, w, c, radius, self.k4_flags,
                None, _stream()), meta)
            _abi.check(code, 'um_local_corr_with_flow_feat')
            return out
        code = self._launch('local_corr_with_flow', lambda: self.lib.um_local_corr_with_flow(
            _ptr(f0), _ptr(f1), _ptr(flow), _ptr(out), b, h, w, c, radius, _stream()), meta)
        _abi.check(code, 'um_local_corr_with_flow')
        return out

    def prop_local(self, q, k, value, h, w, radius):
        b, l, c = q.shape
        _check_tokens('q', q, tokens=h * w)
        _check_tokens('k', k, b, l)
        _check_map('value', value, b, h, w)
        out = torch.empty_like(value)
        code = self._launch('prop_local', lambda: self.lib.um_prop_local_attn(
            _ptr(q), _ptr(k), _ptr(value), _ptr(out), b, h, w, c, value.shape[1], radius, _stream()))
        _abi.check(code, 'um_prop_local_attn')
        return out

    def depth_corr_softmax(self, f0, f1, h, w, cam, candidates, from_argmax=False):
        """cam ``[B, 30]`` = K^-1 | R | t | K (row major), candidates ``[D]`` inverse depths."""
        b, l, c = f0.shape
        _check_tokens('f0', f0, tokens=h * w)
        _check_tokens('f1', f1, b, l)
        if not (cam.is_cuda and cam.dtype == torch.float32 and cam.is_contiguous() and tuple(cam.shape) == (b, 30)):
            raise ValueError(f'cam: expected contiguous CUDA float32 [{b}, 30]')
        if not (candidates.is_cuda and candidates.dtype == torch.float32 and candidates.is_contiguous()
                and candidates.dim() == 1):
            raise ValueError('candidates: expected contiguous CUDA float32 [D]')
        out = torch.empty((b, 1, h, w), dtype=torch.float32, device=f0.device)
        code = self._launch('depth_corr_softmax', lambda: self.lib.um_depth_corr_softmax(
            _ptr(f0), _ptr(f1), _ptr(cam), _ptr(candidates), _ptr(out), b, h, w, c,
            candidates.numel(), int(bool(from_argmax)), _stream()))
        _abi.check(code, 'um_depth_corr_softmax')
        return out
