#!/bin/bash
# Round 3, GPU call E: software-pipelined one-wave-per-SIMD attention (libpipe.so) -- tests and same-box A/B against head.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd "$R"
export TMPDIR=/tmp
V=unimatch_amd/_variants
L=${1:-pipe}
UM_LIB=$V/lib$L.so timeout 300 python -m pytest tests/test_hip_parity_gpu.py -m gpu -x -q -k "projection or fused_layer" > "$OUT/e_pytest_$L.log" 2>&1; echo "pytest rc=$?" >> "$OUT/e_pytest_$L.log"
tail -15 "$OUT/e_pytest_$L.log" | cut -c1-200
timeout 300 python tools/ab_bench.py --steps 30 head= $L=UM_LIB=$V/lib$L.so > "$OUT/e_ab_$L.txt" 2>&1
cat "$OUT/e_ab_$L.txt"
echo "== quantization $L"; UM_LIB=$V/lib$L.so timeout 200 python tools/attn_quantization.py 2>&1 | grep streams | tee "$OUT/e_quant_$L.txt"
C=/tmp/um_parity_cache
UM_LIB=$V/lib$L.so timeout 600 python tools/parity_fullsize.py --configs 2 --weights ctor326,conditioned --seeds 1 --kinds shift --cache $C > "$OUT/e_parity_$L.txt" 2>&1
grep -E "ALL" "$OUT/e_parity_$L.txt" | cut -c1-200
