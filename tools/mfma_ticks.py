"""Diagnostic (GPU box): calibrate s_memtime against the matrix pipe (um_debug_mfma_ticks) -- tick rate under load and ticks per
v_mfma_f32_32x32x16_f16 for one / two waves per SIMD, constant / pseudo-random operands, one accumulator / four."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimatch_amd import _abi  # noqa: E402

lib = _abi.load_diagnostic()      # um_debug_*: `python -m unimatch_amd.build --variant diag`, UM_LIB=unimatch_amd/_variants/libdiag.so
sink = torch.zeros(1, device='cuda')
stream = torch.cuda.current_stream().cuda_stream
iters = 40000
for waves in (4, 8):
    for rnd in (0, 1):
        for chain in (0, 1):
            ticks = torch.zeros(256 * waves, dtype=torch.int64, device='cuda')
            lib.um_debug_mfma_ticks(ticks.data_ptr(), sink.data_ptr(), 2000, rnd, waves, chain, stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.um_debug_mfma_ticks(ticks.data_ptr(), sink.data_ptr(), iters, rnd, waves, chain, stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            t = ticks.double().mean().item()
            n = 8.0 * iters
            flops = 256 * waves * n * 32768
            print(f'{waves // 4} wave(s)/SIMD  {"random  " if rnd else "constant"}  {"1 accumulator " if chain else "4 accumulators"}: '
                  f'{ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  tick rate {t / ms / 1e6:5.2f} GHz  {t / n:6.2f} ticks per MFMA and wave  '
                  f'{ms * 1e6 / n * (4.0 / waves) * (waves / 4):6.2f} ns per MFMA and wave', flush=True)

for mode, what in ((0, 'A fragments from LDS (QK^T pattern)'), (1, 'same stream, MFMAs on fixed A registers')):
    tiles = 20000
    ticks = torch.zeros(1024, dtype=torch.int64, device='cuda')
    lib.um_debug_mfma_lds(ticks.data_ptr(), sink.data_ptr(), 500, mode, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.um_debug_mfma_lds(ticks.data_ptr(), sink.data_ptr(), tiles, mode, stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    t = ticks.double().mean().item()
    n = 24.0 * tiles
    print(f'1 wave(s)/SIMD  LDS-fed   {what}: {ms:8.3f} ms  {1024 * n * 32768 / ms / 1e9:7.1f} TFLOP/s  tick rate {t / ms / 1e6:5.2f} GHz  '
          f'{t / n:6.2f} ticks per MFMA and wave', flush=True)
