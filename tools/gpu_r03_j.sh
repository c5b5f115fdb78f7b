#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
timeout 120 python tools/mfma_ticks.py 2>&1 | grep "wave" | tee "$OUT/j_mfma_ticks.txt"
timeout 300 python -m pytest tests/test_hip_parity_gpu.py -m gpu -x -q -k "glue or depth_path or end_to_end or graph" 2>&1 | tail -3
timeout 420 python tools/parity_fullsize.py --configs 3,4 --weights conditioned --seeds 3 --kinds shift,noise --fast \
    --cache gpurun_cache/parity --out "$OUT/j_parity_cfg34.json" > "$OUT/j_parity_cfg34.txt" 2>&1; echo "rc=$?" >> "$OUT/j_parity_cfg34.txt"
grep -E "cfg|rc=" "$OUT/j_parity_cfg34.txt" | cut -c1-200
timeout 420 python tools/parity_fullsize.py --configs 5 --weights conditioned --seeds 3 --kinds noise \
    --out "$OUT/j_parity_cfg5n.json" > "$OUT/j_parity_cfg5n.txt" 2>&1; echo "rc=$?" >> "$OUT/j_parity_cfg5n.txt"
grep -E "cfg|rc=" "$OUT/j_parity_cfg5n.txt" | cut -c1-200
