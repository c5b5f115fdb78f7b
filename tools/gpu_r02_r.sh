#!/bin/bash
# round-2 GPU call R: LDS-DMA without the M0 save / restore in every MFMA kernel -- full GPU suite + same-box ABAB
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -6 > "$OUT/r_gpu_tests.log"
timeout 600 python tools/ab_bench.py --steps 30 old=UM_LIB=unimatch_amd/_variants/libold.so new= 2>&1 | tail -4 > "$OUT/r_ab.log"
tail -3 "$OUT/r_gpu_tests.log"; cat "$OUT/r_ab.log"
