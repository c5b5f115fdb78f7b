"""Diagnostic (GPU box): sustained rate of a memory-free MFMA loop on this part (um_debug_mfma_peak)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unimatch_amd import _abi  # noqa: E402

lib = _abi.load_diagnostic()      # um_debug_*: `python -m unimatch_amd.build --variant diag`, UM_LIB=unimatch_amd/_variants/libdiag.so
sink = torch.zeros(1, device='cuda')
stream = torch.cuda.current_stream().cuda_stream
for rnd, iters in ((0, 20000), (0, 100000), (1, 20000), (1, 100000)):
    lib.um_debug_mfma_peak(sink.data_ptr(), 200, rnd, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.um_debug_mfma_peak(sink.data_ptr(), iters, rnd, stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    flops = 1024 * 8 * (8.0 * iters) * (32 * 32 * 16 * 2)
    print(f'{"random  " if rnd else "constant"} operands, iters {iters:7d}: {ms:9.3f} ms  {flops / ms / 1e9:9.1f} TFLOP/s  ({100 * flops / ms / 1e9 / 2500:.1f} % of 2.5 PFLOP/s)', flush=True)
