#!/bin/bash
# round-2 GPU call I: cost volume on the matrix cores (coherent-flow tiles) -- tests and A/B against the VALU kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "cost_volume or local_kernels or weight_range" 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -25 > "$OUT/i_tests.log"
{ for m in 0 1; do echo "== UM_K4_MFMA=$m"; UM_K4_MFMA=$m timeout 120 python tools/bench_ops.py local --iters 20 2>&1 | grep "cost volume"; done
  echo "== UM_K4_MFMA=1 UM_K4_FORCE_VALU=1"; UM_K4_FORCE_VALU=1 timeout 120 python tools/bench_ops.py local --iters 20 2>&1 | grep "cost volume"; } > "$OUT/i_bench.log" 2>&1
tail -12 "$OUT/i_tests.log"; cat "$OUT/i_bench.log"
