#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
V=unimatch_amd/_variants
L=${1:-pipe}
UM_LIB=$V/lib$L.so timeout 300 python -m pytest tests/test_hip_parity_gpu.py -m gpu -x -q -k "projection or fused_layer" 2>&1 | tail -4 | cut -c1-200
echo "== trace"; UM_LIB=$V/libtrace$L.so timeout 120 python tools/trace_attn.py A,B,C,wait,barrier 2>&1 | grep "^wg" | head -8 | tee "$OUT/g_trace_$L.txt"
echo "== quantization $L"; UM_LIB=$V/lib$L.so timeout 200 python tools/attn_quantization.py 2>&1 | grep -E "streams  (16|32)" | tee "$OUT/g_quant_$L.txt"
echo "== quantization head"; timeout 200 python tools/attn_quantization.py 2>&1 | grep -E "streams  (16|32)"
