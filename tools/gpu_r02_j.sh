#!/bin/bash
# round-2 GPU call J: full GPU suite with the matrix-core cost volume; configs 3 / 4 with random-like and conditioned weights, A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -15 > "$OUT/j_gpu_tests.log"
{ for wts in damped conditioned; do for m in 0 1; do echo "== weights=$wts UM_K4_MFMA=$m"; UM_K4_MFMA=$m timeout 200 python tools/bench_configs.py --only 3,4 --steps 10 --weights $wts 2>&1 | grep cfg; done; done; } > "$OUT/j_configs.log" 2>&1
tail -4 "$OUT/j_gpu_tests.log"; cat "$OUT/j_configs.log"
