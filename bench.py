#!/usr/bin/env python3
"""Throughput bench of the UniMatch global-matching hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): GMFlow scale-1 optical flow, batch 8 image pairs of 512x768 per GPU,
synthetic frames, seeded random-init weights.  A "step" is one full forward of the drop-in ``UniMatch`` module
(CNN encoder -> 6-block swin Transformer -> global correlation softmax -> self-attention propagation -> convex
upsampling) with the inputs already resident in HBM; with N GPUs every rank runs its own batch (weak scaling,
no data-path collective) and the per-rank predictions are all-gathered over RCCL at the end of each step.
Rank 0 prints ONE JSON line; ``value`` is whole-job image-pairs/s.

Extra objects on the line:
  roofline      the dominant HIP kernel (windowed attention): algorithmic FLOPs per launch
                / its mean launch duration, measured with hipEvents recorded on the launch stream inside
                the timed region, against the dense 16-bit MFMA peak.
  cpu_baseline  the CPU port (oracle/, a torch-CPU restatement of the reference pinned to it by golden
                fixtures; /root/reference does not exist on the GPU box) timed on a bounded sample of the
                same workload on this box's host cores; also yields the EPE delta of the GPU output.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_16BIT = 2.5e15        # dense bf16/fp16 MFMA peak of MI355X (MI355X_MICROARCH.md)
HEIGHT, WIDTH, BATCH = 512, 768, 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--precision', default='exact', choices=['exact', 'fast'],
                    help="'exact' (default, parity mode: fp16 hi+lo split MFMA operands) or 'fast' (bf16 operands)")
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-iters', type=int, default=8)
    return ap.parse_args()


UM_K_COUNT = 12      # include/unimatch_hip.h
# what a memory-free MFMA loop with pseudo-random operands sustains on an MI355X under its power limit
# (tools/mfma_peak.py, profiles/r01_mfma_sustained_peak.txt); the data-sheet peak is only reached with constant operands
SUSTAINED_MFMA = 1.72e15


def collect(lib, kid):
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    lib.um_timing_collect(kid, ctypes.byref(ms), ctypes.byref(n))
    return ms.value, n.value


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    import torch.distributed as dist
    # UM_BENCH_FORCE_DIST=1 exercises the RCCL path (process group + all-gather) even with a single rank
    distributed = world > 1 or os.environ.get('UM_BENCH_FORCE_DIST') == '1'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)           # "nccl" is RCCL on ROCm

    from unimatch_amd import UniMatch, _abi
    from unimatch_amd.synth import CONFIGS, synth_images, synth_state_dict
    ck, fk = CONFIGS['gmflow_s1']
    model = UniMatch(**ck).eval()
    sd = synth_state_dict({k: v.shape for k, v in model.state_dict().items()})
    model.load_state_dict(sd)
    model = model.to(dev).set_precision(args.precision)
    lib = _abi.load()

    b = args.batch
    # distinct frames per rank (seeded), resident in HBM before the timed region
    i0, i1 = synth_images(b, HEIGHT, WIDTH, seed=1000 + rank, kind='shift')
    i0, i1 = i0.to(dev), i1.to(dev)
    # The RCCL all-gather of step k (25 MB per rank at config 2) runs on RCCL's own stream while step k+1 computes: two
    # receive buffers, the handle of the previous gather is waited for before the next one is issued and after the last step
    # (inside the timed region), so K timed steps contain K complete all-gathers.
    gathered = [torch.empty(world * b, 2, HEIGHT, WIDTH, device=dev) for _ in range(2)] if distributed else None
    pending = {'work': None, 'src': None, 'i': 0}

    def finish_gather():
        if pending['work'] is not None:
            pending['work'].wait()
            pending['work'] = pending['src'] = None

    def step():
        pred = model(i0, i1, **fk)['flow_preds'][0]
        if distributed:
            finish_gather()
            src = pred.contiguous()
            pending['work'] = dist.all_gather_into_tensor(gathered[pending['i'] & 1], src, async_op=True)
            pending['src'] = src                       # keep the send buffer alive until the collective has completed
            pending['i'] += 1
        return pred

    for _ in range(args.warmup):
        pred = step()
    finish_gather()
    torch.cuda.synchronize()
    lib.um_timing_enable((1 << 0) | (1 << 1))       # only the kernels the roofline blocks report: window_attn, gsv
    for kid in range(UM_K_COUNT):
        collect(lib, kid)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pred = step()
    finish_gather()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lib.um_timing_enable(0)
    if distributed:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    attn_ms, attn_n = collect(lib, 0)
    gsv_ms, gsv_n = collect(lib, 1)
    split_ms, split_n = collect(lib, 2)

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    # ---- algorithmic work: feature map 64x96, L=6144, C=128, K=2 -> n=1536, 2B streams
    h, w, c = HEIGHT // 8, WIDTH // 8, 128
    L, n = h * w, (h // 2) * (w // 2)
    attn_flops = 4.0 * (2 * b) * L * n * c                          # QK^T + PV per launch
    if getattr(model.ops, 'fused_merge', False):
        attn_flops += 2.0 * (2 * b) * L * c * c                     # + the merge Linear folded into the epilogue
    gsv_flops = b * (2.0 * L * L * c + 4.0 * L * L)                 # per launch (corr or propagation)
    issued = 3.0 if args.precision == 'exact' else 1.0
    # HBM traffic of the dominant kernel: PMC counters are collected in separate rocprofv3 passes (they cannot be
    # read from inside this process); the corrected per-launch figure for this exact launch shape is kept in
    # profiles/ (see the note inside the file) and quoted here when the shape matches.
    traffic = gsv_traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_final.json')))
        if b == BATCH and args.precision == 'exact':
            traffic = round(pm['window_attn_kernel<Fp16, 2, true>']['hbm_traffic_bytes_per_launch'] / 1e6, 1)
            gsv_traffic = round(pm['gsv_kernel<Fp16, 2, 2, false>']['hbm_traffic_bytes_per_launch'] / 1e6, 1)
    except (OSError, KeyError, ValueError):
        pass
    roof = None
    if attn_n:
        dur = attn_ms / attn_n * 1e-3
        ach = attn_flops / dur
        roof = {'kernel': 'window_attn_kernel', 'bound': 'mfma', 'achieved': round(ach / 1e12, 2),
                'peak': PEAK_MFMA_16BIT / 1e12, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_MFMA_16BIT, 4),
                'traffic': traffic, 'traffic_unit': 'MB per launch (rocprofv3 PMC, profiles/r01_pmc_final.json)',
                'launches': attn_n, 'avg_launch_ms': round(attn_ms / attn_n, 4),
                'algorithmic_gflop_per_launch': round(attn_flops / 1e9, 2),
                'issued_mfma_frac': round(ach * issued / PEAK_MFMA_16BIT, 4),
                'sustained_mfma_peak_measured': SUSTAINED_MFMA / 1e12,
                'issued_frac_of_sustained': round(ach * issued / SUSTAINED_MFMA, 4)}
    roof2 = None
    if gsv_n:
        dur = gsv_ms / gsv_n * 1e-3
        ach = gsv_flops / dur
        roof2 = {'kernel': 'gsv_kernel (global correlation / propagation)', 'bound': 'mfma',
                 'achieved': round(ach / 1e12, 2), 'peak': PEAK_MFMA_16BIT / 1e12, 'unit': 'TFLOP/s',
                 'frac': round(ach / PEAK_MFMA_16BIT, 4), 'traffic': gsv_traffic, 'launches': gsv_n,
                 'avg_launch_ms': round(gsv_ms / gsv_n, 4), 'issued_mfma_frac': round(ach * issued / PEAK_MFMA_16BIT, 4)}

    # ---- CPU baseline: the pinned port of the reference, bounded sample, same workload shape
    cpu = None
    epe = {}
    if not args.no_cpu_baseline and world == 1 and rank == 0:      # reported baseline: rank 0 of the 1-GPU run only
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
               'gpu_vs_cpu_fp32': round(_epe(g, ref), 6),
               'note': 'mean end-point error in pixels at full resolution on 1 sample pair; the middle figure is '
                       'the fp32 reference-port noise floor at random-init weights'}

    pairs = world * b * args.steps
    value = pairs / elapsed
    line = {
        'metric': 'image_pairs_per_sec', 'value': round(value, 3), 'unit': 'pairs/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f16x2' if args.precision == 'exact' else 'bf16',
        'data': 'synthetic',
        'config': {'workload': f'GMFlow scale-1 flow, batch {b} x {HEIGHT}x{WIDTH} per GPU, swin K=2, global '
                               'correlation + global propagation, random-init weights',
                   'per_gpu_batch': b, 'global_batch': world * b, 'precision': args.precision,
                   'precision_note': 'exact = fp16 hi+lo split MFMA operands (3 products), fp32 accumulate/softmax; '
                                     'Transformer linears / LayerNorm / FFN, encoder and mask-head convolutions on the same split-fp16 MFMA '
                                     'kernels (no MIOpen kernel in the forward; two small hipBLASLt GEMMs remain: the propagation layer\'s Linear with bias)',
                   'parallelism': f'dp{world} (batch-sharded, all-gather of predictions)' if distributed else 'single GPU'},
        'roofline': roof, 'roofline_global_corr': roof2,
        'split_planes_ms_per_step': round(split_ms / args.steps, 3) if split_n else None,
        'cpu_baseline': cpu, 'epe': epe or None,
        'speedup_vs_cpu_port': None if cpu is None else round(value / cpu['value'], 1),
    }
    print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
