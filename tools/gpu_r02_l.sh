#!/bin/bash
# round-2 GPU call L: local correlation softmax on the matrix-core path -- tests, A/B, configs 3 / 4
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "local_corr or local_kernels or cost_volume or e2e or end_to_end" 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -12 > "$OUT/l_tests.log"
{ for m in 0 1; do echo "== UM_K4_MFMA=$m"; UM_K4_MFMA=$m timeout 120 python tools/bench_ops.py local --iters 20 2>&1 | grep "local corr softmax\|cost volume"; done
  for m in 0 1; do echo "== UM_K4_MFMA=$m"; UM_K4_MFMA=$m timeout 200 python tools/bench_configs.py --only 3,4 --steps 10 2>&1 | grep cfg; done; } > "$OUT/l_bench.log" 2>&1
tail -5 "$OUT/l_tests.log"; cat "$OUT/l_bench.log"
