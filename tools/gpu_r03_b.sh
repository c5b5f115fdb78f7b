#!/bin/bash
# Round 3, GPU call B: balanced attention launch plan (whole rounds + key-split remainder round) -- tests and same-box A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd "$R"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -m gpu -x -q -k "attention or transformer or projection or split_handoffs or graph or fused_layer or end_to_end" > "$OUT/b_pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/b_pytest.log"
tail -4 "$OUT/b_pytest.log"
V=unimatch_amd/_variants
timeout 500 python tools/ab_bench.py --steps 30 nobal=UM_LIB=$V/libdbg.so,UM_WATTN_NO_BALANCE=1 bal=UM_LIB=$V/libdbg.so gm_r02=UM_LIB=$V/libgm_r02.so head= > "$OUT/b_ab.txt" 2>&1
cat "$OUT/b_ab.txt"
timeout 200 python tools/attn_quantization.py > "$OUT/b_quant.txt" 2>&1; cat "$OUT/b_quant.txt"
