#!/bin/bash
# round-2 GPU call O: window_attn_kernel micro-optimisations -- tests + A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "attention or fused_layer or transformer or scale_sweep" 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -6 > "$OUT/o_tests.log"
{ for hr in 0 8; do echo "== exact UM_WATTN_HEADROOM=$hr"; UM_WATTN_HEADROOM=$hr timeout 120 python tools/bench_ops.py attn --precision exact --iters 20 2>&1 | grep "attn"; done; } > "$OUT/o_bench.log" 2>&1
tail -3 "$OUT/o_tests.log"; cat "$OUT/o_bench.log"
