#!/bin/bash
# round-2 GPU call S: attention tests + same-box ABAB (old = previous commit's library in unimatch_amd/_variants/libold.so)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "attention or fused_layer or transformer or scale_sweep or e2e or end_to_end" 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -6 > "$OUT/s_tests.log"
timeout 600 python tools/ab_bench.py --steps 30 old=UM_LIB=unimatch_amd/_variants/libold.so new= 2>&1 | tail -4 > "$OUT/s_ab.log"
tail -3 "$OUT/s_tests.log"; cat "$OUT/s_ab.log"
