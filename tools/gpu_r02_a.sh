#!/bin/bash
# round-2 GPU call A: gsv3 correctness + A/B, full-size parity table, stereo anomaly stage errors, bench line, RCCL world-1.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -x -q -k "global_matching or propagation or scale_sweep" > "$OUT/a_gsv_tests.log" 2>&1
if ! grep -q " passed" "$OUT/a_gsv_tests.log" || grep -q "failed" "$OUT/a_gsv_tests.log"; then
  echo "gsv3 FAILED its tests: the rest of this call runs the round-1 kernel (UM_GSV_V2=1)" | tee "$OUT/a_note.txt"
  GSV=1
else GSV=0; fi
{ for prec in exact fast; do for v2 in 0 1; do echo "== precision=$prec UM_GSV_V2=$v2"; UM_GSV_V2=$v2 timeout 120 python tools/bench_ops.py gsv --precision $prec --iters 20; done; done
  echo "== attention fast"; timeout 120 python tools/bench_ops.py attn --precision fast --iters 10; } > "$OUT/a_ops_bench.log" 2>&1
export UM_GSV_V2=$GSV
timeout 600 python tools/parity_fullsize.py --fast --out "$OUT/r02_parity.json" > "$OUT/r02_parity_fullsize.txt" 2>&1
for wts in damped random conditioned; do
  timeout 300 python tests/diagnostics/stage_error.py --config gmstereo_s2_rr3 --size 384 1248 --weights $wts --variants default,miopen
done > "$OUT/r02_stage_error_stereo_384x1248.txt" 2>&1
timeout 400 python bench.py > "$OUT/a_bench.json" 2> "$OUT/a_bench.err"
UM_BENCH_FORCE_DIST=1 timeout 200 python bench.py --no-cpu-baseline --no-fast --steps 5 --warmup 2 > "$OUT/a_bench_dist1.json" 2> "$OUT/a_bench_dist1.err"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/a_gpu_tests.log" 2>&1
tail -3 "$OUT/a_gsv_tests.log" "$OUT/a_gpu_tests.log"; cat "$OUT/a_ops_bench.log" | grep -v Warn | tail -20; tail -c 600 "$OUT/a_bench.json"
