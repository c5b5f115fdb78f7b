#!/bin/bash
# Round 3, GPU call C: 8-wave / 256-query attention workgroups (libw8.so) -- tests and same-box A/B against head.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd "$R"
export TMPDIR=/tmp
V=unimatch_amd/_variants
UM_LIB=$V/libw8.so timeout 600 python -m pytest tests/test_hip_parity_gpu.py -m gpu -x -q -k "attention or transformer or projection or fused_layer or end_to_end or graph" > "$OUT/c_pytest_w8.log" 2>&1; echo "pytest rc=$?" >> "$OUT/c_pytest_w8.log"
tail -4 "$OUT/c_pytest_w8.log"
timeout 300 python -m pytest tests/test_hip_parity_gpu.py -m gpu -x -q -k "projection" > "$OUT/c_pytest_head.log" 2>&1; tail -2 "$OUT/c_pytest_head.log"
timeout 400 python tools/ab_bench.py --steps 30 head= w8=UM_LIB=$V/libw8.so > "$OUT/c_ab.txt" 2>&1
cat "$OUT/c_ab.txt"
echo "== quantization head"; timeout 200 python tools/attn_quantization.py 2>&1 | grep streams | tee "$OUT/c_quant_head.txt"
echo "== quantization w8"; UM_LIB=$V/libw8.so timeout 200 python tools/attn_quantization.py 2>&1 | grep streams | tee "$OUT/c_quant_w8.txt"
C=/tmp/um_parity_cache
UM_LIB=$V/libw8.so timeout 600 python tools/parity_fullsize.py --configs 2 --weights ctor326,conditioned --seeds 1 --kinds shift --cache $C > "$OUT/c_parity_w8.txt" 2>&1
grep -E "ALL" "$OUT/c_parity_w8.txt"
