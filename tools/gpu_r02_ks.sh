#!/bin/bash
# round-2 GPU call KS: key-split attention for small launches -- tests, batch-1 latency with / without, headline ABAB vs the previous library
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "attention or query_projection or fused_layer or transformer or end_to_end" 2>&1 | grep -v "Warn\|amdgpu.ids" | grep -v "^$" | tail -25 > "$OUT/ks_tests.log"
echo "== key split on" > "$OUT/ks_lat.log"; timeout 200 python tools/bench_graph.py 2>&1 | grep "B=" | cut -c1-60 >> "$OUT/ks_lat.log"
echo "== UM_WATTN_NO_KSPLIT=1" >> "$OUT/ks_lat.log"; UM_WATTN_NO_KSPLIT=1 timeout 200 python tools/bench_graph.py 2>&1 | grep "B=" | cut -c1-60 >> "$OUT/ks_lat.log"
timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fast 2>/dev/null | tail -1 | cut -c1-140 > "$OUT/ks_ab.log"
tail -3 "$OUT/ks_tests.log"; cat "$OUT/ks_lat.log" "$OUT/ks_ab.log"
