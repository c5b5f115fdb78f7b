#!/usr/bin/env python3
"""Throughput bench of the UniMatch global-matching hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: either started by a launcher (``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``: RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or, when run as plain ``python``, this script launches its
own N ranks through ``torch.distributed.run``.  It never measures fewer GPUs than asked for: a node with fewer than N
GPUs, or a launcher whose WORLD_SIZE differs from --gpus, is an error.

    python bench.py --workload cfg4 --gpus N      BASELINE.json configs[3] as written: GMFlow scale-2 + 6 refinements, GLOBAL batch 32
                                                  sharded over the N ranks through ShardedUniMatch + um_allgather_preds (strong scaling)

Workload (default; BASELINE.json configs[1]): GMFlow scale-1 optical flow, batch 8 image pairs of 512x768 per GPU, synthetic frames,
seeded random-init weights.  A "step" is one full forward of the drop-in ``UniMatch`` module (CNN encoder -> 6-block swin
Transformer -> global correlation softmax -> self-attention propagation -> convex upsampling) with the inputs already
resident in HBM; with N GPUs every rank runs its own batch (weak scaling, no data-path collective) and the per-rank
predictions are all-gathered over RCCL (``um_allgather_preds`` of the library's C ABI = ncclAllGather, on a side stream
so that the gather of step k overlaps step k+1) -- K timed steps contain K complete all-gathers.
Rank 0 prints ONE JSON line; ``value`` is whole-job image-pairs/s in EXACT mode (the parity mode).
The step is ``model(img0, img1, ...)`` exactly as a caller of the reference writes it (evaluate_flow.py:405-412).  Round 6: ``UniMatch.forward``
itself decides how the batch is launched (``unimatch_amd.streams.forward_parts``, a pure function of the call): at this workload two concurrent forwards
of 4 pairs on two HIP streams -- the samples of a batch are independent, and the halves fill one another's launch tails; every half is bitwise the plain
forward of its samples.  ``--streams N`` forces a count (1 = one forward of 8); the line carries the one-forward figure as ``serial`` either way.

The headline region is EVENT-FREE (round 4): ``value`` / ``ms_per_step`` come from K steps bracketed by barrier + synchronize with
no per-kernel event inside.  Round 6: that region is run three times back to back and the MEDIAN region is reported (``regions``,
``region_ms_per_step_min_max``).  Directly after it a BREAKDOWN pass runs the same K steps twice with the library's per-kernel hipEvents
(``um_timing_*``, recorded on the launch stream): once timing only the launches inside the CNN encoder, once only those outside it.

Extra objects on the line:
  serial                throughput of the same K steps as ONE forward of the per-GPU batch per step (``config.streams`` = 1 when the
                        headline itself ran that way): the mode of every per-kernel figure below -- in the concurrent mode the
                        durations of kernels that share the GPU overlap and do not price a kernel.
  box                   what THIS box sustains, measured in this run by the library's probe kernels (um_probe_*): a memory-free random-operand
                        fp16 MFMA loop (``mfma_sustained_tflops``), a float4 copy (``hbm_copy_tbps``), a dependent-load chase, back-to-back launches.
                        Boxes of the pool differ by several per cent; two lines differ in ``box`` by what they differ in ``value``.
  roofline              the dominant HIP kernel (windowed attention): algorithmic FLOPs per launch / its mean
                        launch duration from the breakdown pass, against the dense 16-bit MFMA peak; ``traffic`` / ``mfma_busy`` /
                        ``lds_busy`` / ``valu_busy`` come from rocprofv3 PMC passes kept in profiles/ and are nulled when the kernel
                        source has changed since those passes.  ``frac`` prices attention proper (4 L n C per stream); the merge Linear and query
                        projection the same launch executes are in ``with_fused_linears``; ``masked_tile_skip`` = the key-tile census of one
                        forward (what the launches executed of the algorithmic FLOPs: wholly masked tiles of the shifted-window launches are
                        probed and dropped under a bound -- ``frac`` stays on the FULL algorithmic FLOPs).
  roofline_global_corr  the same for ``gsv4_kernel`` (global correlation / propagation: the kernel the north star names).
  roofline_ffn          the same for ``ffn_kernel`` (the whole Transformer FFN in one launch, transformer.py:141-144).
  hot_path_ms_per_step  sum of the kernel durations of everything OUTSIDE the CNN encoder (SURVEY.md 8's path: Transformer,
                        matching, propagation, upsampling head / refinement), per step, with ``hot_path_kernels`` = the split by
                        kernel id; ``encoder_ms_per_step`` = the same sum over the encoder's launches (out of
                        scope, but 45 % of the step); both from the breakdown pass, never from the headline region.
  fast                  the bf16 throughput mode of the same workload (pairs/s, both rooflines, EPE vs fp64): reported beside
                        the headline, never as the headline and never as a parity claim.
  cpu_baseline          the CPU port (oracle/, a torch-CPU restatement of the reference pinned to it by golden fixtures;
                        /root/reference does not exist on the GPU box) timed on a bounded sample of the same workload on
                        this box's host cores; also yields the EPE delta of the GPU output.
  rocm_eager_baseline   the same port executed with stock PyTorch-ROCm eager ops on this GPU (what running the reference's
                        own Python on the MI355X gives: hipBLASLt matmuls, materialised L x L softmax, MIOpen convolutions).
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_16BIT = 2.5e15        # dense bf16/fp16 MFMA peak of MI355X (MI355X_MICROARCH.md)
HEIGHT, WIDTH, BATCH = 512, 768, 8
UM_K_COUNT = 12                 # include/unimatch_hip.h
REGIONS = 3                     # the K-step headline region is repeated this many times inside the run; `value` is the MEDIAN region
PMC_FILE = os.path.join(ROOT, 'profiles', 'pmc_current.json')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--precision', default='exact', choices=['exact', 'fast'],
                    help="headline mode: 'exact' (default, parity mode: fp16 hi+lo split MFMA operands) or 'fast' (bf16)")
    ap.add_argument('--workload', default='cfg2', choices=['cfg2', 'cfg4'],
                    help="cfg2 (default): GMFlow-s1, 8 pairs per GPU, weak scaling.  cfg4: GMFlow-s2 + 6 refinements, GLOBAL batch 32 "
                         "sharded over the ranks (BASELINE.json configs[3] as written), strong scaling")
    ap.add_argument('--batch', type=int, default=None, help='pairs per GPU (cfg2, default 8) / global batch (cfg4, default 32)')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU / ROCm-eager baselines and the EPE legs')
    ap.add_argument('--no-fast', action='store_true', help='skip the extra bf16-mode measurement')
    ap.add_argument('--streams', type=int, default=None,
                    help='default: the step is model(img0, img1, ...) as a caller of the reference would write it -- UniMatch.forward picks '
                         'the number of concurrent forwards itself (unimatch_amd.streams.forward_parts: 2 at configs 2 / 4, the halves fill '
                         'one another\'s launch tails).  N: force N concurrent forwards (1 = one forward of the whole batch).  The roofline / '
                         'breakdown pass always runs one forward with its launches serialised, and the line carries the one-forward '
                         'throughput as `serial`')
    ap.add_argument('--graph', action='store_true',
                    help='replay the HIP graph of the forward (unimatch_amd.graph) in the timed steps instead of launching eagerly '
                         '(measured on MI355X at config 2: 817.5 / 817.0 pairs/s against 818.6 / 819.1 eager -- the step is GPU-bound, '
                         'the default stays eager)')
    ap.add_argument('--cpu-iters', type=int, default=8)
    ap.add_argument('--set', action='append', default=[], metavar='Class.attr=value',
                    help='A/B knob for tools/ab_bench.py: set a class attribute of HipOps / CNNEncoder before the run, e.g. '
                         'HipOps.fused_merge=0 (the product reads no environment variable)')
    return ap.parse_args()


def apply_knobs(specs):
    from unimatch_amd.encoder import CNNEncoder
    from unimatch_amd.ops import HipOps
    classes = {'HipOps': HipOps, 'CNNEncoder': CNNEncoder}
    for spec in specs:
        target, _, value = spec.partition('=')
        cls, _, attr = target.partition('.')
        if cls not in classes or not hasattr(classes[cls], attr):
            raise SystemExit(f'bench.py --set: unknown knob {target}')
        old = getattr(classes[cls], attr)
        setattr(classes[cls], attr, type(old)(int(value)) if isinstance(old, (bool, int)) else type(old)(value))


def collect(lib, kid):
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    lib.um_timing_collect(kid, ctypes.byref(ms), ctypes.byref(n))
    return ms.value, n.value


def source_stamp(files):
    """sha256 over the kernel sources a PMC traffic figure depends on (profiles/pmc_current.json carries the same stamp)."""
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, 'unimatch_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_key, files):
    """Per-launch HBM traffic (MB) from the tracked rocprofv3 PMC pass, or None when absent / taken on other kernel code."""
    try:
        pm = json.load(open(PMC_FILE))
        rec = pm[kernel_key]
        if rec.get('source_stamp') != source_stamp(files):
            return None, 'stale: kernel source changed since the PMC pass'
        return round(rec['hbm_traffic_bytes_per_launch'] / 1e6, 1), f"rocprofv3 --pmc pass at {pm.get('git', '?')}"
    except (OSError, KeyError, ValueError):
        return None, 'no PMC pass on record for this launch shape'


def pmc_mfma_busy(kernel_key, files):
    """Matrix-pipe busy fraction of the kernel from the tracked PMC pass: SQ_VALU_MFMA_BUSY_CYCLES (= 32 cycles per 32x32x16
    MFMA, summed over all SIMDs) / (1024 SIMDs x kernel cycles, kernel cycles = GRBM_GUI_ACTIVE summed over the 8 XCDs / 8).
    Source-stamped like the traffic figure: None when the kernel source changed since the pass."""
    try:
        pm = json.load(open(PMC_FILE))
        rec = pm[kernel_key]
        if rec.get('source_stamp') != source_stamp(files) or 'mfma_busy' not in rec:
            return None, 'stale or absent: no SQ pass on record for this kernel source'
        return rec['mfma_busy'], (f"SQ_VALU_MFMA_BUSY_CYCLES {rec['SQ_VALU_MFMA_BUSY_CYCLES']:.0f} / (1024 SIMDs x GRBM_GUI_ACTIVE "
                                  f"{rec['GRBM_GUI_ACTIVE']:.0f} / 8 XCDs), rocprofv3 --pmc pass at {pm.get('git', '?')}")
    except (OSError, KeyError, ValueError):
        return None, 'no PMC pass on record for this launch shape'


def pmc_other_units(kernel_key, files):
    """LDS-array and VALU busy fractions of the kernel from the same tracked PMC passes (SQ_LDS_IDX_ACTIVE / (256 CUs x cycles);
    4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles), MFMA issue included): what the matrix pipe is NOT waiting for.  Source-stamped."""
    try:
        rec = json.load(open(PMC_FILE))[kernel_key]
        if rec.get('source_stamp') != source_stamp(files):
            return None, None
        return rec.get('lds_busy'), rec.get('valu_busy')
    except (OSError, KeyError, ValueError):
        return None, None


def box_probe(lib, dev):
    """What THIS box sustains, measured in this run (csrc/probe.hip): a ~25 ms memory-free MFMA loop with pseudo-random operands, a
    float4 copy of 1 GiB (x 8), and a dependent-load chase.  MI355X boxes of the pool differ by several per cent; a reader can
    normalise `value` and the roofline fractions by these."""
    stream = torch.cuda.current_stream(dev).cuda_stream
    sink = torch.zeros(1, device=dev)
    iters = 20000

    def ev():
        return torch.cuda.Event(enable_timing=True)
    lib.um_probe_mfma(sink.data_ptr(), 500, stream)
    a, b_ = ev(), ev()
    a.record()
    lib.um_probe_mfma(sink.data_ptr(), iters, stream)
    b_.record()
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).fill_(1)
    dst = torch.empty_like(src)
    lib.um_probe_copy(src.data_ptr(), dst.data_ptr(), nbytes, stream)
    c, d = ev(), ev()
    c.record()
    for _ in range(8):
        lib.um_probe_copy(src.data_ptr(), dst.data_ptr(), nbytes, stream)
    d.record()
    # pointer ring: 1 Mi entries spread over 256 MiB (one entry per 256 bytes), visiting order a fixed odd-multiplier permutation
    n = 1 << 20
    idx = (torch.arange(n, dtype=torch.int64) * 741103597 + 12345) % n            # visiting order (odd multiplier: a permutation)
    nxt = torch.empty(n, dtype=torch.int64)
    nxt[idx] = idx.roll(-1)
    ring = torch.zeros(n * 64, dtype=torch.int32)
    ring[::64] = (nxt * 64).to(torch.int32)
    ring = ring.to(dev)
    out = torch.zeros(1, dtype=torch.int32, device=dev)
    hops = 20000
    lib.um_probe_chase(ring.data_ptr(), out.data_ptr(), 100, stream)
    e, f = ev(), ev()
    e.record()
    lib.um_probe_chase(ring.data_ptr(), out.data_ptr(), hops, stream)
    f.record()
    # launch latency: 200 launches of the smallest kernel at hand
    lib.um_probe_chase(ring.data_ptr(), out.data_ptr(), 1, stream)
    g, h = ev(), ev()
    g.record()
    for _ in range(200):
        lib.um_probe_chase(ring.data_ptr(), out.data_ptr(), 1, stream)
    h.record()
    torch.cuda.synchronize(dev)
    mf = lib.um_probe_mfma_flops(iters) / (a.elapsed_time(b_) * 1e-3)
    return {'mfma_sustained_tflops': round(mf / 1e12, 1), 'mfma_sustained_frac_of_peak': round(mf / PEAK_MFMA_16BIT, 4),
            'hbm_copy_tbps': round(8 * 2 * nbytes / (c.elapsed_time(d) * 1e-3) / 1e12, 3),
            'dependent_load_ns': round(e.elapsed_time(f) * 1e6 / hops, 1),
            'back_to_back_launch_us': round(g.elapsed_time(h) * 1e3 / 200, 2),
            'device': torch.cuda.get_device_name(dev),
            'note': 'measured in this run (um_probe_*): random-operand fp16 MFMA loop on every SIMD (~25 ms), 8 float4 copies of 1 GiB '
                    '(read + write counted), 20000 dependent loads over 256 MiB, 200 back-to-back single-wave launches'}


def roofline_block(name, key, files, timing, flops_per_step, launches_per_step, precision, pmc_ok, bound='mfma', sustained=None):
    """One roofline object: `achieved` = algorithmic FLOPs of the kernel's launches in a step / the summed hipEvent
    duration of those launches (breakdown pass).  With one launch shape per step this is per launch; with several (config 4: two
    scales) it is the time-weighted aggregate and `avg_launch_ms` the plain mean."""
    ms, n = timing
    if not n:
        return None
    issued = 3.0 if precision == 'exact' else 1.0
    steps = n / launches_per_step
    dur = ms * 1e-3 / steps                      # seconds of this kernel per step
    ach = flops_per_step / dur
    traffic, note = pmc_traffic(key, files) if pmc_ok else (None, 'no PMC pass for this workload / batch')
    busy, busy_note = pmc_mfma_busy(key, files) if pmc_ok else (None, 'no PMC pass for this workload / batch')
    lds_busy, valu_busy = pmc_other_units(key, files) if pmc_ok else (None, None)
    return {'kernel': name, 'bound': bound, 'achieved': round(ach / 1e12, 2), 'peak': PEAK_MFMA_16BIT / 1e12,
            'unit': 'TFLOP/s', 'frac': round(ach / PEAK_MFMA_16BIT, 4), 'traffic': traffic,
            'traffic_unit': 'MB per launch', 'traffic_source': note, 'launches': n,
            'avg_launch_ms': round(ms / n, 4), 'algorithmic_gflop_per_launch': round(flops_per_step / launches_per_step / 1e9, 2),
            'issued_mfma_frac': round(ach * issued / PEAK_MFMA_16BIT, 4),
            'mfma_busy': busy, 'mfma_busy_source': busy_note, 'lds_busy': lds_busy, 'valu_busy': valu_busy,
            'issued_frac_of_box_sustained': (round(ach * issued / sustained, 4) if sustained else None),
            'duration_source': 'hipEvent pairs on the launch stream (um_timing_*), breakdown pass after the event-free headline region'}


def cpu_model_string():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


KERNEL_NAMES = ['window_attn', 'global_softmax (gsv3/gsv4)', 'split_planes', 'local_corr_softmax', 'cost_volume (K4)', 'prop_local',
                'depth_corr', 'linear', 'instance_norm / nhwc', 'convex_upsample / warp', 'ffn', 'conv']


def run_as_launcher(argv, gpus, need_gpus=True):
    """`python bench.py --gpus N` without a launcher: start the N ranks and pass rank 0's line on ONLY when every rank exited 0.
    A job in which any rank failed prints a line with ``value: null``, the exit code and the tail of the ranks' stderr instead --
    a failed rank can never produce a number.  Returns the exit code."""
    from unimatch_amd.dist import launch_ranks
    rc, out, err = launch_ranks(os.path.abspath(__file__), argv, gpus, need_gpus=need_gpus, capture=True)
    sys.stderr.write(err)
    if rc != 0:
        print(json.dumps({'metric': 'image-pairs/sec', 'value': None, 'unit': 'pairs/s', 'n_gpus': gpus, 'higher_is_better': True,
                          'error': f'a rank of the {gpus}-rank job exited with code {rc}: no measurement', 'exit_code': rc,
                          'stderr_tail': err[-6000:]}), flush=True)
        return rc
    sys.stdout.write(out)
    sys.stdout.flush()
    return 0


def main():
    args = parse()
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher (fails loudly when the node has fewer than N GPUs)
        sys.exit(run_as_launcher(sys.argv[1:], args.gpus))
    world = int(env_world or '1')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit(f'bench.py: rank {rank} needs GPU {local_rank}; this node exposes '
                         f'{torch.cuda.device_count() if torch.cuda.is_available() else 0}')
    import torch.distributed as dist
    # UM_BENCH_FORCE_DIST=1 exercises the RCCL path (process group + communicator + all-gather) even with a single rank
    distributed = world > 1 or os.environ.get('UM_BENCH_FORCE_DIST') == '1'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    cfg4 = args.workload == 'cfg4'
    gather, gather_kind = None, None
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)           # "nccl" is RCCL on ROCm: launcher-side barrier / reductions
        from unimatch_amd import dist as umd
        if cfg4:                                                  # ShardedUniMatch gathers through the process-wide collective
            gather = umd.rccl_gather(dev)
            gather_kind = umd.GATHER_KIND
        else:
            gather, gather_kind = umd.make_gather(rank, world, dev, id_file=os.environ.get('UM_RCCL_ID_FILE'))

    from unimatch_amd import UniMatch, _abi
    from unimatch_amd.dist import ShardedUniMatch, shard_bounds
    from unimatch_amd.synth import CONFIGS, synth_images, synth_state_dict
    apply_knobs(args.set)
    cfg_name = 'gmflow_s2_rr6' if cfg4 else 'gmflow_s1'
    ck, fk = CONFIGS[cfg_name]
    model = UniMatch(**ck).eval()
    sd = synth_state_dict({k: v.shape for k, v in model.state_dict().items()})
    model.load_state_dict(sd)
    model = model.to(dev)
    lib = _abi.load()

    if cfg4:
        gb = args.batch or 32                                     # GLOBAL batch, sharded over the ranks (strong scaling)
        lo, hi = shard_bounds(gb, rank, world)
        b = hi - lo                                               # pairs this rank computes
        i0, i1 = synth_images(gb, HEIGHT, WIDTH, seed=1000, kind='shift')      # every rank holds the full batch, as a caller would
        runner = ShardedUniMatch(model, rank=rank, world=world, force_gather=distributed)
    else:
        b = args.batch or BATCH
        gb = world * b
        # distinct frames per rank (seeded), resident in HBM before the timed region
        i0, i1 = synth_images(b, HEIGHT, WIDTH, seed=1000 + rank, kind='shift')
    i0, i1 = i0.to(dev), i1.to(dev)
    # cfg2: the all-gather of step k (25 MB per rank) runs on a side stream while step k+1 computes: two receive
    # buffers; the compute stream waits for the previous gather before the next one is issued and after the last step
    # (inside the timed region), so K timed steps contain K complete all-gathers.  cfg4: ShardedUniMatch gathers on the compute stream.
    side = torch.cuda.Stream(device=dev) if distributed and not cfg4 else None
    gathered = [torch.empty(world, b, 2, HEIGHT, WIDTH, device=dev) for _ in range(2)] if side is not None else None
    pending = {'event': None, 'src': None, 'i': 0}

    def finish_gather():
        if pending['event'] is not None:
            torch.cuda.current_stream(dev).wait_event(pending['event'])
            pending['event'] = pending['src'] = None

    # --graph: the timed steps replay the HIP graph of the forward (one capture per precision, bitwise equal to the eager launches:
    # tests/test_hip_parity_gpu.py::test_hip_graph_replay_matches_eager); the default and the breakdown pass launch eagerly.
    launch = {'fwd': model, 'mode': 'eager'}

    def step():
        if cfg4:
            runner.model = launch['fwd']
            return runner(i0, i1, **fk)['flow_preds'][0]          # [global batch, 2, H, W] on every rank
        pred = launch['fwd'](i0, i1, **fk)['flow_preds'][0]
        if distributed:
            finish_gather()
            src = pred.contiguous()
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))
            side.wait_event(ready)
            gather.all_gather(src, gathered[pending['i'] & 1], stream=side)
            done = torch.cuda.Event()
            done.record(side)
            pending['event'], pending['src'] = done, src      # keep the send buffer alive until the collective completed
            pending['i'] += 1
        return pred

    step_ms = []

    concurrent = {}

    region_spread = []

    def timed(precision, steps, warmup, streams=None, regions=1):
        """The headline region: K steps, barrier + synchronize on both sides, no per-kernel event inside."""
        model.set_precision(precision)
        launch['fwd'], launch['mode'] = model, 'eager'
        streams = args.streams if streams is None else streams
        model.launch_parts = streams                                  # None: UniMatch.forward's own plan (the drop-in call)
        from unimatch_amd.streams import forward_parts
        nparts = min(b, streams if streams is not None else forward_parts(fk['task'], fk['attn_type'], ck['num_scales'], ck['reg_refine'],
                                                                           b, HEIGHT, WIDTH))
        concurrent['parts'] = nparts
        if nparts > 1:
            launch['mode'] = (f'eager, model(img0, img1, ...): {nparts} concurrent forwards of {b // nparts} pairs on {nparts} HIP streams '
                              + ('(forced by --streams)' if streams is not None else '(UniMatch.forward\'s own plan, streams.forward_parts)'))
        if args.graph:
            from unimatch_amd.graph import GraphedUniMatch
            launch['fwd'], launch['mode'] = GraphedUniMatch(model), 'hip_graph_replay'      # captured by the first warm-up step
        for _ in range(max(warmup, 2)):                              # (the stream wrapper's first call of a geometry is sequential)
            pred = step()
        if launch['mode'].startswith('hip_graph') and any(v is False for v in launch['fwd']._graphs.values()):
            launch['fwd'], launch['mode'] = model, 'eager (HIP graph capture failed)'
        finish_gather()
        torch.cuda.synchronize()
        lib.um_timing_enable(0)
        # The region -- EXACTLY K steps bracketed by barrier + synchronize on both sides, max over ranks -- is run `regions` times back
        # to back and the MEDIAN region is reported (round 6: one 0.18 s region is at the mercy of the box's clock ramp; min / max of
        # the regions are on the line as `region_ms_per_step_min_max`).
        results = []
        for _ in range(regions):
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            # per-step device time for the MEDIAN step: one event between steps on the compute stream, read after the
            # timed region (no host synchronisation inside it); `value` stays K steps / wall time of the bracketed region (the contract)
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            t0 = time.perf_counter()
            marks[0].record()
            for k in range(steps):
                pred = step()
                marks[k + 1].record()
            finish_gather()
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            if distributed:
                tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                elapsed = tmax.item()
            results.append((elapsed, sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(steps))))
        results.sort(key=lambda r: r[0])
        elapsed, steps_sorted = results[len(results) // 2]
        step_ms.clear()
        step_ms.extend(steps_sorted)
        region_spread.clear()
        region_spread.extend([results[0][0] / steps * 1e3, results[-1][0] / steps * 1e3])
        mode_used = launch['mode']
        launch['fwd'], launch['mode'] = model, 'eager'             # the breakdown pass needs the library's per-launch events
        model.launch_parts = 1                                      # ... of ONE forward with its launches serialised
        return elapsed, pred, mode_used

    def breakdown(steps):
        """After the headline region: the same steps with the library's per-kernel hipEvents -- one pass timing only the launches
        INSIDE the CNN encoder, one pass timing only those outside it (hooks on the backbone flip the timing mask; nothing is
        synchronised inside a step).  Returns ({kernel id: (ms, launches)} outside, the same inside, encoder wall ms per step)."""
        out = []
        spans = []
        for inside in (False, True):
            state = {'e0': None}

            def pre(_m, _a, inside=inside, state=state):
                lib.um_timing_enable(-1 if inside else 0)
                if inside:
                    state['e0'] = torch.cuda.Event(enable_timing=True)
                    state['e0'].record()

            def post(_m, _a, _o, inside=inside, state=state):
                lib.um_timing_enable(0 if inside else -1)
                if inside:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    spans.append((state['e0'], e1))
            h1 = model.backbone.register_forward_pre_hook(pre)
            h2 = model.backbone.register_forward_hook(post)
            torch.cuda.synchronize()
            for kid in range(UM_K_COUNT):
                collect(lib, kid)
            lib.um_timing_enable(0 if inside else -1)
            for _ in range(steps):
                step()
            finish_gather()
            torch.cuda.synchronize()
            lib.um_timing_enable(0)
            out.append({kid: collect(lib, kid) for kid in range(UM_K_COUNT)})
            h1.remove()
            h2.remove()
        enc_wall = sum(a.elapsed_time(b_) for a, b_ in spans) / max(len(spans), 1)
        return out[0], out[1], enc_wall

    elapsed, pred, launch_mode = timed(args.precision, args.steps, args.warmup, regions=REGIONS)
    headline_parts = concurrent['parts']
    headline_regions = [round(v, 3) for v in region_spread]
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    spread_ms = (step_ms[0], step_ms[-1])
    serial = None
    if 'concurrent' in launch_mode:                                # the same K steps as ONE forward per step (what the breakdown pass times)
        s_el, _, _ = timed(args.precision, args.steps, 2, streams=1)
        serial = {'value': round(world * b * args.steps / s_el, 3), 'unit': 'pairs/s', 'ms_per_step': round(s_el / args.steps * 1e3, 3),
                  'note': 'one forward of the whole per-GPU batch per step, launches serialised on one stream: the mode the roofline '
                          'durations, hot_path_ms_per_step and encoder_ms_per_step of this line are measured in'}
    hot_t, enc_t, enc_wall_ms = breakdown(args.steps)
    # key-tile census of the attention launches of ONE forward (um_window_attn_tile_census): what the masked-tile skip executed
    from unimatch_amd import _abi as _abi_mod
    model.launch_parts = 1
    _abi_mod.attn_tile_census(True)
    step()
    finish_gather()
    tile_census = _abi_mod.attn_tile_census(False)
    other = 'fast' if args.precision == 'exact' else 'exact'
    extra = None
    if not args.no_fast:
        e_el, e_pred, _ = timed(other, args.steps, max(2, args.warmup // 2))
        extra = (e_el, e_pred) + breakdown(args.steps)
        model.set_precision(args.precision)
    rccl_ranks = gather.ranks() if gather is not None else 1

    if rank != 0:
        if distributed:
            gather.close()
            dist.destroy_process_group()
        return
    box = box_probe(lib, dev)
    sustained = box['mfma_sustained_tflops'] * 1e12

    # ---- algorithmic work per step on THIS rank: per scale, feature map h x w, L = h w, C = 128, K splits -> n = L / K^2,
    # S = 2 b streams; 12 attention launches, 6 FFN launches per scale; global correlation + global propagation at scale 0
    c = 128
    scales = [(HEIGHT // 8, WIDTH // 8, 2)] if not cfg4 else [(HEIGHT // 8, WIDTH // 8, 2), (HEIGHT // 4, WIDTH // 4, 8)]
    S = 2 * b
    attn_flops = sum(12 * 4.0 * S * (h * w) * ((h // k) * (w // k)) * c for h, w, k in scales)
    attn_fused = 0.0
    if getattr(model.ops, 'fused_merge', False):
        attn_fused += sum(12 * 2.0 * S * h * w * c * c for h, w, _ in scales)       # the merge Linear folded into the epilogue
        if getattr(model.ops, 'fused_qproj', False):
            attn_fused += sum(12 * 2.0 * S * h * w * c * c for h, w, _ in scales)   # the query projection folded into the prologue
    L0 = scales[0][0] * scales[0][1]
    gsv_flops = 2 * b * (2.0 * L0 * L0 * c + 4.0 * L0 * L0)       # correlation + propagation launch at scale 0
    ffn_flops = sum(6 * 2.0 * S * h * w * 1024 * 384 for h, w, _ in scales)
    n_attn, n_gsv, n_ffn = 12 * len(scales), 2, 6 * len(scales)

    def blocks(hot, precision):
        from unimatch_amd.ops import HipOps
        tag = 'Fp16, 2' if precision == 'exact' else 'Bf16, 1'
        pmc_ok = (not cfg4) and b == BATCH
        f_, r_, k_ = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib.um_window_attn_plan(S, HEIGHT // 8, WIDTH // 8, HEIGHT // 16, WIDTH // 16, ctypes.byref(f_), ctypes.byref(r_), ctypes.byref(k_))
        ksplit = 'true' if r_.value > 0 else 'false'
        r1 = roofline_block('window_attn_kernel', f"window_attn_kernel<{tag}, true, {'true' if HipOps.fused_qproj else 'false'}, {ksplit}>",
                            ['window_attn.hip', 'common.h'], hot[0], attn_flops, n_attn, precision, pmc_ok, sustained=sustained)
        if r1 is not None and attn_fused:
            tot = (attn_flops + attn_fused) / (hot[0][0] * 1e-3 / (hot[0][1] / n_attn))
            issued = 3.0 if precision == 'exact' else 1.0
            r1['with_fused_linears'] = {
                'note': 'the same launches also execute transformer.py:58 (q projection, prologue) and :137 (merge Linear, '
                        'epilogue; LayerNorm + residual not counted)',
                'algorithmic_gflop_per_launch': round((attn_flops + attn_fused) / n_attn / 1e9, 2), 'achieved': round(tot / 1e12, 2),
                'frac': round(tot / PEAK_MFMA_16BIT, 4), 'issued_mfma_frac': round(tot * issued / PEAK_MFMA_16BIT, 4)}
        r2 = roofline_block('gsv4_kernel (global correlation / propagation)', f'gsv4_kernel<{tag}, 2>', ['global_match.hip', 'common.h'],
                            hot[1], gsv_flops, n_gsv, precision, pmc_ok, sustained=sustained)
        fused_kv = all(getattr(HipOps, k_, False) for k_ in ('block_kv', 'fused_kv', 'fused_ffn', 'fused_qproj', 'fused_merge'))
        # with fused_kv every FFN launch but the last block's also executes the NEXT block's four k | v projections (um_ffn_kv_fwd):
        # `frac` prices what the launches execute (FFN + those projections); the FFN proper is in `ffn_only`
        kv_flops = sum(5 * 2.0 * S * h * w * 512 * 128 for h, w, _ in scales) if fused_kv else 0.0
        r3 = roofline_block('ffn_kernel (whole Transformer FFN' + (' + next block\'s k|v projections' if fused_kv else '') + ', one launch)',
                            f"ffn_kernel<{tag}, false, {'true' if fused_kv else 'false'}>", ['ffn.hip', 'common.h'],
                            hot[10], ffn_flops + kv_flops, n_ffn, precision, pmc_ok, sustained=sustained)
        if r3 is not None:
            dur = hot[10][0] * 1e-3 / (hot[10][1] / n_ffn)
            r3['ffn_only'] = {'note': 'FFN FLOPs alone (transformer.py:141-144: 2 M 8C 3C per launch) over the same launch time'
                                      + (' -- which 5 of 6 launches share with the k | v projections of the next block' if fused_kv else ''),
                              'algorithmic_gflop_per_launch': round(ffn_flops / n_ffn / 1e9, 2),
                              'frac': round(ffn_flops / dur / PEAK_MFMA_16BIT, 4)}
        return r1, r2, r3

    def sums(hot, enc, steps):
        per = {KERNEL_NAMES[k]: round(v[0] / steps, 4) for k, v in hot.items() if v[1]}
        return (round(sum(v[0] for v in hot.values()) / steps, 3), per, round(sum(v[0] for v in enc.values()) / steps, 3),
                {KERNEL_NAMES[k]: round(v[0] / steps, 4) for k, v in enc.items() if v[1]})

    roof, roof2, roof3 = blocks(hot_t, args.precision)
    if roof is not None and tile_census.get('workgroups'):
        # one (128-query workgroup, 32-key tile) pair = 4 * 128 * 32 * C algorithmic FLOP (QK^T + PV); a probe = the hi.hi third of
        # QK^T = 1/6 of that.  `frac` / `achieved` above stay on SURVEY 8(d)'s FULL 4 L n C: dropped tiles are work not done, not
        # throughput -- this object says how much of the algorithmic work the launches executed.
        unit = 4.0 * 128 * 32 * c
        full, probed, refused = tile_census['full'], tile_census['probed'], tile_census['probed_then_computed']
        roof['executed_gflop_per_launch'] = round((full + probed / 6.0) * unit / n_attn / 1e9, 2)      # (detail: masked_tile_skip)
        roof['masked_tile_skip'] = {
            'tiles_computed': full, 'tiles_probed': probed, 'probed_tiles_computed_after_all': refused,
            'tiles_dropped': probed - refused, 'workgroups': tile_census['workgroups'],
            'executed_gflop_per_launch': round((full + probed / 6.0) * unit / n_attn / 1e9, 2),
            'executed_frac_of_algorithmic': round((full + probed / 6.0) / max(full + probed - refused, 1), 4),
            'note': 'key-tile census of the attention launches of one forward (um_window_attn_tile_census): wholly masked (workgroup, '
                    'key tile) pairs of the shifted-window launches are probed (hi.hi product) and dropped only when every logit stays '
                    '40 natural-log units below the running row maximum (unimatch/attention.py:88-89, unimatch/utils.py:84-108)'}
    hot_ms, hot_per, enc_ms, enc_per = sums(hot_t, enc_t, args.steps)

    # ---- baselines + EPE (rank 0 of the 1-GPU run only): bounded samples of the same workload shape, one pair
    cpu, eager, epe, epe_other = None, None, {}, None
    if not args.no_cpu_baseline and world == 1:
        from oracle import model as om
        c0, c1 = i0[:1].cpu(), i1[:1].cpu()
        okw = dict(fk, num_scales=ck['num_scales'], upsample_factor=ck['upsample_factor'], reg_refine=ck['reg_refine'])
        ncores = os.cpu_count() or 1
        # torch's CPU ops stop scaling (and then collapse) long before 256 threads: pick the fastest of a
        # short sweep, then time the bounded sample with it and report the thread count actually used
        best_t, best_s, ref = None, None, None
        for threads in [t for t in (8, 16, 32, 64) if t <= ncores] or [ncores]:
            torch.set_num_threads(threads)
            if ref is None:
                ref = om.unimatch_forward(sd, c0, c1, **okw)         # warm-up + fp32 parity sample
            t1 = time.perf_counter()
            om.unimatch_forward(sd, c0, c1, **okw)
            dt = time.perf_counter() - t1
            if best_s is None or dt < best_s:
                best_t, best_s = threads, dt
        torch.set_num_threads(best_t)
        t1 = time.perf_counter()
        iters = 0
        while iters < args.cpu_iters and (time.perf_counter() - t1) < 25.0:
            om.unimatch_forward(sd, c0, c1, **okw)
            iters += 1
        cpu_s = (time.perf_counter() - t1) / max(iters, 1)
        cpu = {'value': round(1.0 / cpu_s, 4), 'unit': 'pairs/s', 'cores': best_t, 'kind': 'port',
               'host_cores': ncores, 'cpu_model': cpu_model_string(), 'torch_threads': torch.get_num_threads(),
               'sample': f'{iters} forwards of 1 pair {HEIGHT}x{WIDTH} (fp32 torch-CPU port of the reference, '
                         f'pinned to it by tests/golden), {cpu_s:.2f} s each'}
        truth = om.unimatch_forward(sd, c0.double(), c1.double(), **okw)     # fp64 evaluation = ground truth

        def _epe(a, b_):
            return (a.double() - b_.double()).pow(2).sum(1).sqrt().mean().item()
        g = pred[:1].cpu()
        epe = {'gpu_vs_fp64_truth': round(_epe(g, truth), 6), 'cpu_fp32_vs_fp64_truth': round(_epe(ref, truth), 6),
               'gpu_vs_cpu_fp32': round(_epe(g, ref), 6), 'precision': args.precision,
               'note': 'mean end-point error in pixels at full resolution on 1 sample pair; the middle figure is '
                       'the fp32 reference-port noise floor at random-init weights.  Every sample of the batch, both image '
                       'kinds, 3 seeds, all five configs at their own batch: profiles/r03_parity_batch.txt; configs 3 / 4 (chaotic '
                       'end to end at random init) stage by stage on the reference constructor\'s weights: profiles/r04_stage_parity.txt; configs 1 / 2 / 5 stage by stage: '
                       'profiles/r05_stage_parity_one_scale.txt'}
        if extra is not None:
            epe_other = round(_epe(extra[1][:1].cpu(), truth), 6)
        # the same port on this GPU with stock PyTorch-ROCm eager ops (factories default to the device inside the context)
        try:
            sd_dev = {k: v.to(dev) for k, v in sd.items()}
            eb = min(b, 2)
            with torch.device(dev):
                for _ in range(2):
                    om.unimatch_forward(sd_dev, i0[:eb], i1[:eb], **okw)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                eit = 5
                for _ in range(eit):
                    eo = om.unimatch_forward(sd_dev, i0[:eb], i1[:eb], **okw)
                torch.cuda.synchronize()
            es = (time.perf_counter() - t1) / eit
            eager = {'value': round(eb / es, 3), 'unit': 'pairs/s', 'kind': 'port on PyTorch-ROCm eager (fp32, hipBLASLt / MIOpen / ATen)',
                     'sample': f'{eit} forwards of {eb} pairs {HEIGHT}x{WIDTH}, {es * 1e3:.1f} ms each',
                     'epe_vs_fp64_truth': round(_epe(eo[:1].cpu(), truth), 6)}
            del sd_dev, eo
        except Exception as exc:      # a baseline, not the product: report why it is missing instead of failing the bench
            eager = {'value': None, 'error': f'{type(exc).__name__}: {exc}'[:200]}

    fast_obj = None
    if extra is not None:
        e_el, _, e_hot, e_enc, _ = extra
        r1, r2, r3 = blocks(e_hot, other)
        e_hot_ms, _, e_enc_ms, _ = sums(e_hot, e_enc, args.steps)
        fast_obj = {'precision': other, 'dtype': 'bf16' if other == 'fast' else 'f16x2',
                    'value': round(gb * args.steps / e_el, 3), 'unit': 'pairs/s',
                    'ms_per_step': round(e_el / args.steps * 1e3, 3), 'roofline': r1, 'roofline_global_corr': r2, 'roofline_ffn': r3,
                    'hot_path_ms_per_step': e_hot_ms, 'encoder_ms_per_step': e_enc_ms,
                    'epe_vs_fp64_truth': epe_other,
                    'note': 'same workload and steps in the other operand precision; reported beside the headline, not a parity claim'}

    pairs = gb * args.steps
    value = pairs / elapsed
    if cfg4:
        workload = (f'GMFlow scale-2 + 6 refinements (BASELINE.json configs[3]), GLOBAL batch {gb} x {HEIGHT}x{WIDTH} sharded over '
                    f'{world} rank(s) ({b} pairs on rank 0), swin K=[2,8], global + local (r=4) correlation, global + local propagation, '
                    'random-init weights')
    else:
        workload = (f'GMFlow scale-1 flow, batch {b} x {HEIGHT}x{WIDTH} per GPU, swin K=2, global '
                    'correlation + global propagation, random-init weights')
    line = {
        'metric': 'image_pairs_per_sec', 'value': round(value, 3), 'unit': 'pairs/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'ms_per_step_median': round(median_ms, 3), 'ms_per_step_min_max': [round(spread_ms[0], 3), round(spread_ms[1], 3)],
        'regions': REGIONS, 'region_ms_per_step_min_max': headline_regions,
        'box': box,
        'pairs_per_sec_at_median': round((gb if cfg4 else b) / (median_ms * 1e-3), 2),
        'higher_is_better': True, 'scaling': 'strong' if cfg4 else 'weak', 'vs_baseline': None,
        'dtype': 'f16x2' if args.precision == 'exact' else 'bf16',
        'data': 'synthetic', 'rccl_ranks': rccl_ranks, 'collective': gather_kind if distributed else None,
        'config': {'workload': workload,
                   'per_gpu_batch': b, 'global_batch': gb, 'precision': args.precision, 'launch_mode': launch_mode,
                   'streams': headline_parts,
                   'precision_note': 'exact = fp16 hi+lo split MFMA operands (3 products), fp32 accumulate/softmax; every '
                                     'GEMM / convolution of the forward runs on the library\'s own split-fp16 MFMA kernels '
                                     '(no MIOpen, hipBLASLt or rocBLAS kernel in the forward)',
                   'parallelism': ((f'dp{world} (batch-sharded through ShardedUniMatch, um_allgather_preds = ncclAllGather of the predictions on the compute stream)'
                                    if cfg4 else
                                    f'dp{world} (batch-sharded, um_allgather_preds = ncclAllGather of predictions, side stream)')
                                   if distributed else 'single GPU'),
                   'weights': 'synth_state_dict(seed 326): per-parameter seeded generator with the reference initialisers\' '
                              'statistics (xavier-uniform / kaiming-normal), rebuilt identically on any box'},
        'roofline': roof, 'roofline_global_corr': roof2, 'roofline_ffn': roof3,
        'hot_path_ms_per_step': hot_ms, 'encoder_ms_per_step': enc_ms,
        'encoder_wall_ms_per_step': round(enc_wall_ms, 3),
        'hot_path_kernels_ms_per_step': hot_per, 'encoder_kernels_ms_per_step': enc_per,
        'untimed_ms_per_step': round((serial['ms_per_step'] if serial else median_ms) - hot_ms - enc_ms, 3),
        'serial': serial,
        'hot_path_pairs_per_sec': round((b if not cfg4 else b) / (hot_ms * 1e-3), 1) if hot_ms else None,
        'timing_note': 'value / ms_per_step: event-free region (barrier + synchronize around K steps), eager launches (--graph replays '
                       'the forward as a HIP graph: no gain, the step is GPU-bound; config.launch_mode).  The step is model(...): UniMatch.forward\'s own '
                       'plan (or --streams N) computes the per-GPU batch as N concurrent forwards on N HIP streams (unimatch_amd.streams: the parts fill one '
                       'another\'s launch tails; every part is bitwise the plain forward of its samples) and `serial` is the same K steps as '
                       'one forward per step; `value` is the median of `regions` back-to-back regions of K steps.  roofline durations (kernels alone on the GPU, one forward per step), '
                       'hot_path_ms_per_step (kernel-duration sum of everything outside the CNN encoder = SURVEY 8\'s path) and '
                       'encoder_ms_per_step (the encoder\'s launches, SURVEY 2 #8, out of scope) come from a separate breakdown '
                       'pass of the same K steps with per-kernel hipEvents on the launch stream; hot_path_pairs_per_sec = this '
                       'rank\'s pairs / hot_path_ms_per_step; untimed_ms_per_step = (serial, else median) step - both sums (ATen element-wise ops such '
                       'as the position add, a few sub-10-us glue kernels, launch gaps)',
        'fast' if other == 'fast' else 'exact': fast_obj,
        'cpu_baseline': cpu, 'rocm_eager_baseline': eager, 'epe': epe or None,
        'speedup_vs_cpu_port': None if cpu is None else round(value / cpu['value'], 1),
        'speedup_vs_rocm_eager': None if not eager or not eager.get('value') else round(value / eager['value'], 1),
    }
    if distributed:
        gather.close()
        dist.destroy_process_group()
    # RCCL writes its version banner to C stdout (buffered until exit): flush it first so that the JSON line is the LAST line
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.write(json.dumps(line) + '\n')
    sys.stdout.flush()


if __name__ == '__main__':
    main()
