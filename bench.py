#!/usr/bin/env python3
"""Throughput bench of the UniMatch global-matching hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: either started by a launcher (``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``: RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or, when run as plain ``python``, this script launches its
own N ranks through ``torch.distributed.run``.  It never measures fewer GPUs than asked for: a node with fewer than N
GPUs, or a launcher whose WORLD_SIZE differs from --gpus, is an error.

Workload (BASELINE.json configs[1]): GMFlow scale-1 optical flow, batch 8 image pairs of 512x768 per GPU, synthetic frames,
seeded random-init weights.  A "step" is one full forward of the drop-in ``UniMatch`` module (CNN encoder -> 6-block swin
Transformer -> global correlation softmax -> self-attention propagation -> convex upsampling) with the inputs already
resident in HBM; with N GPUs every rank runs its own batch (weak scaling, no data-path collective) and the per-rank
predictions are all-gathered over RCCL (``um_allgather_preds`` of the library's C ABI = ncclAllGather, on a side stream
so that the gather of step k overlaps step k+1) -- K timed steps contain K complete all-gathers.
Rank 0 prints ONE JSON line; ``value`` is whole-job image-pairs/s in EXACT mode (the parity mode).

Extra objects on the line:
  roofline              the dominant HIP kernel (windowed attention): algorithmic FLOPs per launch / its mean
                        launch duration, measured with hipEvents recorded on the launch stream inside the timed region
                        (two event records per timed launch: the headline is slightly pessimistic), against the dense 16-bit
                        MFMA peak; ``traffic`` comes from a rocprofv3 PMC pass kept in profiles/ and is nulled when the
                        kernel source has changed since that pass.  ``frac`` prices attention proper (4 L n C per stream);
                        the merge Linear and query projection the same launch executes are in ``with_fused_linears``.
  roofline_global_corr  the same for ``gsv4_kernel`` (global correlation / propagation: the kernel the north star names).
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
# what a memory-free MFMA loop with pseudo-random operands sustains on an MI355X under its power limit
# (tools/mfma_peak.py, profiles/r01_mfma_sustained_peak.txt); the data-sheet peak is only reached with constant operands
SUSTAINED_MFMA = 1.72e15
PMC_FILE = os.path.join(ROOT, 'profiles', 'pmc_current.json')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--precision', default='exact', choices=['exact', 'fast'],
                    help="headline mode: 'exact' (default, parity mode: fp16 hi+lo split MFMA operands) or 'fast' (bf16)")
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU / ROCm-eager baselines and the EPE legs')
    ap.add_argument('--no-fast', action='store_true', help='skip the extra bf16-mode measurement')
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


def rooflines(attn, gsv, flops_attn, flops_gsv, precision, batch, flops_attn_fused=0.0):
    """`achieved` / `frac` price the SURVEY 8(d) figure of the kernel's function alone (attention: 4 L n C per stream); the
    merge Linear and query projection the attention launch also executes are reported beside it, never inside it."""
    from unimatch_amd.ops import HipOps
    issued = 3.0 if precision == 'exact' else 1.0
    tag = ('Fp16, 2' if precision == 'exact' else 'Bf16, 1')
    out = []
    # the instantiation the bench's attention calls take: key-split remainder round inside the launch or not (um_window_attn_plan)
    from unimatch_amd import _abi
    f_, r_, k_ = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _abi.load().um_window_attn_plan(2 * batch, HEIGHT // 8, WIDTH // 8, HEIGHT // 16, WIDTH // 16, ctypes.byref(f_), ctypes.byref(r_), ctypes.byref(k_))
    ksplit = 'true' if r_.value > 0 else 'false'
    for name, key, files, (ms, n), fl in (
            ('window_attn_kernel', f"window_attn_kernel<{tag}, true, {'true' if HipOps.fused_qproj else 'false'}, {ksplit}>",
             ['window_attn.hip', 'common.h'], attn, flops_attn),
            ('gsv4_kernel (global correlation / propagation)', f'gsv4_kernel<{tag}, 2>', ['global_match.hip', 'common.h'],
             gsv, flops_gsv)):
        if not n:
            out.append(None)
            continue
        dur = ms / n * 1e-3
        ach = fl / dur
        traffic, note = pmc_traffic(key, files) if batch == BATCH else (None, 'non-default batch')
        busy, busy_note = pmc_mfma_busy(key, files) if batch == BATCH else (None, 'non-default batch')
        out.append({'kernel': name, 'bound': 'mfma', 'achieved': round(ach / 1e12, 2), 'peak': PEAK_MFMA_16BIT / 1e12,
                    'unit': 'TFLOP/s', 'frac': round(ach / PEAK_MFMA_16BIT, 4), 'traffic': traffic,
                    'traffic_unit': 'MB per launch', 'traffic_source': note, 'launches': n,
                    'avg_launch_ms': round(ms / n, 4), 'algorithmic_gflop_per_launch': round(fl / 1e9, 2),
                    'issued_mfma_frac': round(ach * issued / PEAK_MFMA_16BIT, 4),
                    'mfma_busy': busy, 'mfma_busy_source': busy_note,
                    'sustained_mfma_peak_measured': SUSTAINED_MFMA / 1e12,
                    'issued_frac_of_sustained': round(ach * issued / SUSTAINED_MFMA, 4)})
        if name == 'window_attn_kernel' and flops_attn_fused:
            tot = (fl + flops_attn_fused) / dur
            out[-1]['with_fused_linears'] = {
                'note': 'the same launches also execute transformer.py:58 (q projection, prologue) and :137 (merge Linear, '
                        'epilogue; LayerNorm + residual not counted)',
                'algorithmic_gflop_per_launch': round((fl + flops_attn_fused) / 1e9, 2), 'achieved': round(tot / 1e12, 2),
                'frac': round(tot / PEAK_MFMA_16BIT, 4), 'issued_mfma_frac': round(tot * issued / PEAK_MFMA_16BIT, 4)}
    return out


def main():
    args = parse()
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher (fails loudly when the node has fewer than N GPUs)
        from unimatch_amd.dist import launch_ranks
        sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
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
    gather, gather_kind = None, None
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)           # "nccl" is RCCL on ROCm: launcher-side barrier / reductions
        from unimatch_amd.dist import make_gather
        gather, gather_kind = make_gather(rank, world, dev, id_file=os.environ.get('UM_RCCL_ID_FILE'))   # the data-path collective

    from unimatch_amd import UniMatch, _abi
    from unimatch_amd.synth import CONFIGS, synth_images, synth_state_dict
    apply_knobs(args.set)
    ck, fk = CONFIGS['gmflow_s1']
    model = UniMatch(**ck).eval()
    sd = synth_state_dict({k: v.shape for k, v in model.state_dict().items()})
    model.load_state_dict(sd)
    model = model.to(dev)
    lib = _abi.load()

    b = args.batch
    # distinct frames per rank (seeded), resident in HBM before the timed region
    i0, i1 = synth_images(b, HEIGHT, WIDTH, seed=1000 + rank, kind='shift')
    i0, i1 = i0.to(dev), i1.to(dev)
    # The all-gather of step k (25 MB per rank at config 2) runs on a side stream while step k+1 computes: two receive
    # buffers; the compute stream waits for the previous gather before the next one is issued and after the last step
    # (inside the timed region), so K timed steps contain K complete all-gathers.
    side = torch.cuda.Stream(device=dev) if distributed else None
    gathered = [torch.empty(world, b, 2, HEIGHT, WIDTH, device=dev) for _ in range(2)] if distributed else None
    pending = {'event': None, 'src': None, 'i': 0}

    def finish_gather():
        if pending['event'] is not None:
            torch.cuda.current_stream(dev).wait_event(pending['event'])
            pending['event'] = pending['src'] = None

    def step():
        pred = model(i0, i1, **fk)['flow_preds'][0]
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

    def timed(precision, steps, warmup):
        model.set_precision(precision)
        for _ in range(warmup):
            pred = step()
        finish_gather()
        torch.cuda.synchronize()
        lib.um_timing_enable((1 << 0) | (1 << 1))       # only the kernels the roofline blocks report: window_attn, gsv
        for kid in range(UM_K_COUNT):
            collect(lib, kid)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        # per-step device time for the MEDIAN: one event between steps on the compute stream, read after the timed
        # region (no host synchronisation inside it); `value` stays K steps / wall time of the bracketed region (the contract)
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
        step_ms.clear()
        step_ms.extend(sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(steps)))
        lib.um_timing_enable(0)
        if distributed:
            tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = tmax.item()
        return elapsed, pred, collect(lib, 0), collect(lib, 1), collect(lib, 2)

    elapsed, pred, attn_t, gsv_t, split_t = timed(args.precision, args.steps, args.warmup)
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    spread_ms = (step_ms[0], step_ms[-1])
    other = 'fast' if args.precision == 'exact' else 'exact'
    extra = None
    if not args.no_fast:
        extra = timed(other, args.steps, max(2, args.warmup // 2))
    rccl_ranks = gather.ranks() if gather is not None else 1

    if rank != 0:
        if distributed:
            gather.close()
            dist.destroy_process_group()
        return

    # ---- algorithmic work: feature map 64x96, L=6144, C=128, K=2 -> n=1536, 2B streams
    h, w, c = HEIGHT // 8, WIDTH // 8, 128
    L, n = h * w, (h // 2) * (w // 2)
    attn_flops = 4.0 * (2 * b) * L * n * c                          # QK^T + PV per launch
    attn_fused = 0.0
    if getattr(model.ops, 'fused_merge', False):
        attn_fused += 2.0 * (2 * b) * L * c * c                     # the merge Linear folded into the epilogue
        if getattr(model.ops, 'fused_qproj', False):
            attn_fused += 2.0 * (2 * b) * L * c * c                 # the query projection folded into the prologue
    gsv_flops = b * (2.0 * L * L * c + 4.0 * L * L)                 # per launch (corr or propagation)
    roof, roof2 = rooflines(attn_t, gsv_t, attn_flops, gsv_flops, args.precision, b, attn_fused)

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
               'host_cores': ncores,
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
                       'kinds, 3 seeds, all five configs at their own batch: profiles/r03_parity_batch.txt (weights: the '
                       'reference constructor under seed 326, and the BUILDER-DEFINED conditioned set synth.CONDITIONED for '
                       'the absolute 1e-3 px gate)'}
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
        e_el, _, e_attn, e_gsv, _ = extra
        r1, r2 = rooflines(e_attn, e_gsv, attn_flops, gsv_flops, other, b, attn_fused)
        fast_obj = {'precision': other, 'dtype': 'bf16' if other == 'fast' else 'f16x2',
                    'value': round(world * b * args.steps / e_el, 3), 'unit': 'pairs/s',
                    'ms_per_step': round(e_el / args.steps * 1e3, 3), 'roofline': r1, 'roofline_global_corr': r2,
                    'epe_vs_fp64_truth': epe_other,
                    'note': 'same workload and steps in the other operand precision; reported beside the headline, not a parity claim'}

    pairs = world * b * args.steps
    value = pairs / elapsed
    line = {
        'metric': 'image_pairs_per_sec', 'value': round(value, 3), 'unit': 'pairs/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'ms_per_step_median': round(median_ms, 3), 'ms_per_step_min_max': [round(spread_ms[0], 3), round(spread_ms[1], 3)],
        'pairs_per_sec_at_median': round(b / (median_ms * 1e-3), 2),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f16x2' if args.precision == 'exact' else 'bf16',
        'data': 'synthetic', 'rccl_ranks': rccl_ranks, 'collective': gather_kind if distributed else None,
        'config': {'workload': f'GMFlow scale-1 flow, batch {b} x {HEIGHT}x{WIDTH} per GPU, swin K=2, global '
                               'correlation + global propagation, random-init weights',
                   'per_gpu_batch': b, 'global_batch': world * b, 'precision': args.precision,
                   'precision_note': 'exact = fp16 hi+lo split MFMA operands (3 products), fp32 accumulate/softmax; every '
                                     'GEMM / convolution of the forward runs on the library\'s own split-fp16 MFMA kernels '
                                     '(no MIOpen, hipBLASLt or rocBLAS kernel in the forward)',
                   'parallelism': (f'dp{world} (batch-sharded, um_allgather_preds = ncclAllGather of predictions, side stream)'
                                   if distributed else 'single GPU'),
                   'weights': 'synth_state_dict(seed 326): per-parameter seeded generator with the reference initialisers\' '
                              'statistics (xavier-uniform / kaiming-normal), rebuilt identically on any box'},
        'roofline': roof, 'roofline_global_corr': roof2,
        'timing_note': 'roofline kernel durations come from hipEvent pairs recorded around each window_attn / gsv launch '
                       'INSIDE the timed steps (28 records per step); value is therefore slightly pessimistic',
        'split_planes_ms_per_step': round(split_t[0] / args.steps, 3) if split_t[1] else None,
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
