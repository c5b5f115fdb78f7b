#!/bin/bash
# round-2 GPU call Y: full GPU suite + ABAB (old = library before the upsampler's channels-last specialisation) + all configs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "Warn\|amdgpu.ids" | grep -v "^$" | tail -15 > "$OUT/y_tests.log"
timeout 400 python tools/ab_bench.py --steps 30 old=UM_LIB=unimatch_amd/_variants/libold.so new= 2>&1 | tail -4 > "$OUT/y_ab.log"
timeout 300 python tools/bench_configs.py --steps 10 2>&1 | grep cfg > "$OUT/y_all_configs.txt"
tail -3 "$OUT/y_tests.log"; cat "$OUT/y_ab.log" "$OUT/y_all_configs.txt"
