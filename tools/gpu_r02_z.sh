#!/bin/bash
# round-2 GPU call Z: XCD-aware grid of linear_kernel -- tests + same-box ABAB against the previous library
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "linear or fused_layer or transformer or attention_merge or propagation or e2e or end_to_end" 2>&1 | grep -v "Warn\|amdgpu.ids" | grep -v "^$" | tail -25 > "$OUT/z_tests.log"
timeout 600 python tools/ab_bench.py --steps 30 old=UM_LIB=unimatch_amd/_variants/libold.so new= 2>&1 | tail -4 > "$OUT/z_ab.log"
tail -3 "$OUT/z_tests.log"; cat "$OUT/z_ab.log"
