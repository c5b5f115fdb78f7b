#!/bin/bash
# round-2 GPU call C (after re-entry): where does the tree stand — gsv3 tests + A/B, all GPU tests, default bench line, world-1 RCCL path.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -k "global_matching or propagation or scale_sweep" 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -15 > "$OUT/c_gsv_tests.log"
{ for prec in exact fast; do for v2 in 0 1; do echo "== precision=$prec UM_GSV_V2=$v2"; UM_GSV_V2=$v2 timeout 120 python tools/bench_ops.py gsv --precision $prec --iters 20 2>&1 | grep -v "Warn\|amdgpu.ids"; done; done
  echo "== attention"; timeout 120 python tools/bench_ops.py attn --precision exact --iters 10 2>&1 | grep -v "Warn\|amdgpu.ids"; timeout 120 python tools/bench_ops.py attn --precision fast --iters 10 2>&1 | grep -v "Warn\|amdgpu.ids"; } > "$OUT/c_ops_bench.log" 2>&1
timeout 500 python bench.py > "$OUT/c_bench.json" 2> "$OUT/c_bench.err"; echo "rc=$?" >> "$OUT/c_bench.err"
UM_BENCH_FORCE_DIST=1 NCCL_DEBUG=WARN timeout 200 python -X faulthandler bench.py --no-cpu-baseline --no-fast --steps 5 --warmup 2 > "$OUT/c_bench_dist1.json" 2> "$OUT/c_bench_dist1.err"; echo "rc=$?" >> "$OUT/c_bench_dist1.err"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -30 > "$OUT/c_gpu_tests.log"
tail -3 "$OUT/c_gsv_tests.log"; cat "$OUT/c_ops_bench.log"; tail -5 "$OUT/c_bench_dist1.err"; tail -5 "$OUT/c_gpu_tests.log"; tail -c 1500 "$OUT/c_bench.json"
