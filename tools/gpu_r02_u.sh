#!/bin/bash
# round-2 GPU call U: query projection in the attention prologue -- tests + same-box ABAB (UM_QPROJ=0 = separate q planes)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "attention or fused_layer or transformer or e2e or end_to_end" 2>&1 | grep -v "Warn\|amdgpu.ids" | grep -v "^$" | tail -25 > "$OUT/u_tests.log"
timeout 600 python tools/ab_bench.py --steps 30 planes=UM_QPROJ=0 qproj= 2>&1 | tail -4 > "$OUT/u_ab.log"
tail -3 "$OUT/u_tests.log"; cat "$OUT/u_ab.log"
