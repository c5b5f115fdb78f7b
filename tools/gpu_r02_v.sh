#!/bin/bash
# round-2 GPU call V: single-piece GELU in ffn_kernel -- tests + same-box ABAB against the previous library
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "ffn or fused_layer or transformer or e2e or end_to_end" 2>&1 | grep -v "Warn\|amdgpu.ids" | grep -v "^$" | tail -25 > "$OUT/v_tests.log"
timeout 600 python tools/ab_bench.py --steps 30 old=UM_LIB=unimatch_amd/_variants/libold.so new= 2>&1 | tail -4 > "$OUT/v_ab.log"
timeout 300 python tools/bench_ops.py ffn 2>&1 | tail -6 > "$OUT/v_ops.log"
tail -3 "$OUT/v_tests.log"; cat "$OUT/v_ab.log" "$OUT/v_ops.log"
