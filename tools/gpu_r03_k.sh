#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
V=unimatch_amd/_variants
timeout 500 python -m pytest tests/test_hip_parity_gpu.py -m gpu -x -q -k "attention or transformer or projection or fused_layer or end_to_end or split_handoffs or graph or reproducible" 2>&1 | tail -4 | cut -c1-200
timeout 300 python tools/ab_bench.py --steps 30 nobal=UM_LIB=$V/libdbg.so,UM_WATTN_NO_BALANCE=1 bal=UM_LIB=$V/libdbg.so head= 2>&1 | tee "$OUT/k_ab.txt"
echo "== quantization"; timeout 200 python tools/attn_quantization.py 2>&1 | grep streams | tee "$OUT/k_quant.txt"
