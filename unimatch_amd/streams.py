"""One batch as several concurrent forwards on separate HIP streams of ONE GPU.

The samples of a batch are independent (InstanceNorm is per sample, every matching layer is per pair: /root/reference/unimatch/unimatch.py
:113-367 has no cross-sample operation), and the hot path's launches do not tile the chip evenly: the attention launch of config 2 is 768
workgroups on 512 resident slots (a half-empty last round at 37 % matrix-pipe busy), the FFN runs three synchronised rounds
whose prologues / epilogues overlap with nothing.  Two forwards of half the batch on two streams fill one another's tails: while one half's
launch drains, the other half's next launch starts.  Measured at config 2 (8 x 512x768): 9.67 -> 9.17 ms per step on one box, 9.20 -> 8.92 on
another (+3 ... +5 %); four forwards of two pairs LOSE (10.4 ms: launches of a quarter of the chip).  Default two parts.  Round 6: this is how ``UniMatch.forward`` itself launches
a batch where :func:`forward_parts` says so -- no wrapper needed; ``ConcurrentUniMatch`` remains as the switch that FORCES a part count.

What it is not: a different computation.  Every part is a plain ``UniMatch.forward`` of its samples; results agree with the one-forward
result to the summation order of the launch-size-dependent decompositions (stream-K split points of the global correlation), i.e. to the
fp32 noise floor the parity tables are measured against, and are bitwise those of forwarding the part alone.
"""
import torch

from .dist import shard_batch, shard_bounds


def forward_parts(task, attn_type, num_scales, reg_refine, batch, height, width):
    """How many concurrent forwards ``UniMatch.forward`` cuts a batch into: a pure function of the call.

    Two where the same-box table shows a gain, one where it shows a loss (``profiles/r06_forward_parts.txt``, MI355X, one box, forced 1
    against forced 2 with the mechanics of :class:`PartRunner`; "size" = batch x height x width in units of 512 x 768 pixels):

        one-scale flow (GMFlow-s1)     512x768: 2 pairs -7 %, 3 +7.5 %, 4 +1 %, 6 +12 %, 8 +4.5 %, 16 +2.5 %
                                       320x448: 2 pairs -36 %, 4 -25 % (size 1.5), 8 +6 % (size 2.9), 16 +11 %
        one-scale depth (480x640)      2 pairs -20 % (size 1.6), 4 +11.6 % (size 3.1), 8 +12 %, 16 +6 %
        one-scale stereo (512x960)     2 pairs -8 % (size 2.5), 4 +-0 (size 5), 8 +3.4 % (size 10)
        flow, 2 scales + 6 refinements 2 pairs +4 %, 4 +7 %, 8 +4 %, 16 +2.8 %, 32 +3.1 %
        stereo, 2 scales + 3 refin.    2 pairs +-0 (size 2.5), 4 +3.9 % (size 5)
        three parts                    never better than two (config 2: 838 against 871 pairs/s; config 5: 1162 against 1181)

    The gain comes from launch tails and per-workgroup fixed costs that a second forward's launches fill; it turns into a loss when
    half a batch no longer fills the chip (small launches take the latency-bound split variants).  All five BASELINE configs except the
    single pair of config 1 run as two parts.  (A first table, measured with a side-stream set per model and ``Tensor.record_stream``
    on the parts' outputs, had shown losses for stereo, depth and mid-size batches: artefacts of those mechanics)"""
    if batch < 2:
        return 1
    size = batch * height * width / float(512 * 768)
    if reg_refine:
        need = 1.9 if task == 'flow' else 4.0
    elif attn_type and '1d' in attn_type:                     # 1-D cross attention (stereo): bandwidth-bound launches need bigger halves
        need = 6.0
    else:
        need = 2.8
    return 2 if size >= need else 1


# side streams for parts 1 .. n-1, ONE set per device for the whole process (round 6: a set per model left every new model with
# fresh streams, and HIP multiplexes streams onto a handful of hardware queues -- the fifth and sixth stream of a process landed on
# the queue of another one and two "concurrent" parts ran one after the other plus the cross-stream waits: config 4 at 26.5 ms
# instead of 21.2 when it ran after another configuration in the same process).  Part 0 runs on the caller's stream.
_SIDE_STREAMS = {}


def _side_streams(dev, count):
    pool = _SIDE_STREAMS.setdefault(str(dev), [])
    while len(pool) < count:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:count]


class PartRunner:
    """Runs ``model._forward_one`` on ``n`` contiguous sample ranges -- part 0 on the caller's stream, the others on the process-wide
    side streams of the device -- joined on the caller's stream.

    The first forward of a geometry / argument set / parameter version runs the parts one after the other on the caller's stream:
    that is the forward that builds the caches later forwards only read (weight planes, position tables; buffers that are written per
    forward are kept per stream) -- concurrent first use would race on them.  The same after anything that makes the backend rebuild
    shared entries (``set_precision``, ``invalidate_weights``, a weight edited in place): the parts only go to separate streams when the
    backend object is the one of the previous forward and that forward built no shared cache entry (``HipOps.cache_generation``)."""

    def __init__(self):
        self._seen = set()
        self._backend = None                     # (id of the model's backend, its cache generation) after the previous forward

    @staticmethod
    def _backend_state(model):
        ops = getattr(model, '_ops', None)                         # (UniMatch creates its backend on first use)
        return (id(ops), getattr(ops, 'cache_generation', None))

    @staticmethod
    def _key(model, n, img0, kw):
        version = sum(p._version for p in model.parameters())
        small = tuple(sorted((k, v if isinstance(v, (int, float, bool, str, type(None))) else
                              tuple(v) if isinstance(v, (list, tuple)) else tuple(v.shape)) for k, v in kw.items()))
        return (n, tuple(img0.shape), str(img0.device), small, version)

    def run(self, model, n, img0, img1, kw):
        batch = img0.shape[0]
        bidir = 2 if (kw.get('pred_bidir_flow') or kw.get('pred_bidir_depth')) else 1

        def part_inputs(r):
            pk = dict(kw)
            for key in ('intrinsics', 'pose'):
                if pk.get(key) is not None:
                    pk[key] = shard_batch(pk[key], r, n)
            return shard_batch(img0, r, n), shard_batch(img1, r, n), pk

        def one(r, a0, a1, pk):
            # buffers a captured graph owns are keyed by (graph token, lane): the parts of one capture must not share arrival counters
            # or activation planes (HipOps._split_workspace)
            ops = model.ops if img0.is_cuda else None
            if ops is not None and hasattr(ops, 'workspace_lane'):
                ops.workspace_lane = r
            try:
                return model._forward_one(a0, a1, **pk)['flow_preds']
            finally:
                if ops is not None and hasattr(ops, 'workspace_lane'):
                    ops.workspace_lane = 0

        key = self._key(model, n, img0, kw)
        concurrent = img0.is_cuda and key in self._seen and self._backend is not None and self._backend == self._backend_state(model)
        outs = []
        if not concurrent:
            for r in range(n):
                outs.append(one(r, *part_inputs(r)))
            self._seen.add(key)
        else:
            dev = img0.device
            cur = torch.cuda.current_stream(dev)
            side = _side_streams(dev, n - 1)
            ins = [part_inputs(r) for r in range(n)]                # sliced on the caller's stream
            for s in side:
                s.wait_stream(cur)                                  # (before part 0 is enqueued: the side parts depend on the inputs only)
            outs = [None] * n
            for r in range(1, n):
                with torch.cuda.stream(side[r - 1]):
                    outs[r] = one(r, *ins[r])
            outs[0] = one(0, *ins[0])                               # part 0 on the caller's stream
            for r in range(1, n):
                cur.wait_stream(side[r - 1])
        self._backend = self._backend_state(model)
        # every prediction of the list: [bidir * b_r, ...] per part -> [bidir * batch, ...] in the reference's [forward; backward] order
        counts = [shard_bounds(batch, r, n)[1] - shard_bounds(batch, r, n)[0] for r in range(n)]
        preds = []
        for i in range(len(outs[0])):
            pieces = [outs[r][i].reshape(bidir, counts[r], *outs[r][i].shape[1:]) for r in range(n)]
            preds.append(torch.cat(pieces, 1).reshape(bidir * batch, *outs[0][i].shape[1:]))
        if concurrent:
            # The side parts' outputs live in the side streams' allocator pools and were consumed (concatenated) on the caller's stream.
            # Instead of Tensor.record_stream -- which parks every such block until an event of the caller's stream has completed and
            # made the caching allocator grow in steady state (1 - 2 hipMalloc of a fresh segment per ten steps, none without it; the one
            # mechanism found for intermittent 25 - 55 ms hiccups) -- the side streams are ordered behind the consumer: whatever runs on
            # them next starts after the concatenation.
            # (Inside a HIP-graph capture a side stream that waits on the capturing stream would have to be joined again before the
            # capture ends: there the blocks are handed to the allocator's own cross-stream bookkeeping instead.)
            if torch.cuda.is_current_stream_capturing():
                for r in range(1, n):
                    for t in outs[r]:
                        t.record_stream(cur)
            else:
                for s in side:
                    s.wait_stream(cur)
        return {'flow_preds': preds}


class ConcurrentUniMatch(torch.nn.Module):
    """``ConcurrentUniMatch(model, parts)(img0, img1, **kw)`` = ``model(img0, img1, **kw)`` with the number of concurrent forwards FORCED
    to ``parts`` (``UniMatch.forward`` otherwise picks it per call, :func:`forward_parts`): A/B timing and tests."""

    def __init__(self, model, parts=2):
        super().__init__()
        if parts < 1:
            raise ValueError('parts must be >= 1')
        self.model = model
        self.parts = parts

    @property
    def _backend(self):
        runner = getattr(self.model, '_runner', None)
        return None if runner is None else runner._backend

    def forward(self, img0, img1, **kw):
        prev = self.model.launch_parts
        self.model.launch_parts = self.parts
        try:
            return self.model(img0, img1, **kw)
        finally:
            self.model.launch_parts = prev
