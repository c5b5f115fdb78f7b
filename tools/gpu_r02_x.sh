#!/bin/bash
# round-2 GPU call X: interleaved channel ownership in the gather kernels -- tests + microbenchmarks old / new + configs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "local or cost or prop or depth or stereo or e2e or end_to_end" 2>&1 | grep -v "Warn\|amdgpu.ids" | grep -v "^$" | tail -25 > "$OUT/x_tests.log"
for lib in unimatch_amd/_variants/libold.so ""; do
  echo "== UM_LIB=$lib" >> "$OUT/x_ops.log"
  UM_LIB=$lib UM_K4_MFMA=0 timeout 300 python tools/bench_ops.py local 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -6 >> "$OUT/x_ops.log"
done
tail -3 "$OUT/x_tests.log"; cat "$OUT/x_ops.log"
