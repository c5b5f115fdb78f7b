#!/bin/bash
# round-2 GPU call H: all GPU tests + the profiles/ evidence set at the gsv4 (stream-K) state
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -15 > "$OUT/h_gpu_tests.log"
bash tools/collect_profiles.sh r02 > "$OUT/h_collect.log" 2>&1
tail -4 "$OUT/h_gpu_tests.log"; cat "$OUT/r02_all_configs.txt"; tail -c 1200 "$OUT/r02_bench.json"
