#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
V=unimatch_amd/_variants
timeout 400 python -m pytest tests/test_hip_parity_gpu.py -m gpu -x -q -k "attention or transformer or projection or fused_layer or end_to_end" 2>&1 | tail -4 | cut -c1-200
timeout 300 python tools/ab_bench.py --steps 30 head0=UM_LIB=$V/libhead0.so new= 2>&1 | tee "$OUT/h_ab.txt"
echo "== trace new"; UM_LIB=$V/libtrace.so timeout 120 python tools/trace_attn.py prep,QK,softmax,PV,wait,barrier 2>&1 | grep "^wg" | awk 'NR%3==1' | tee "$OUT/h_trace_head.txt"
