#!/bin/bash
# Round 3, GPU call D: the full parity table at the BASELINE batch sizes (3 seeds x 2 image kinds, every sample vs fp64) + batch-1 variants
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd "$R"
export TMPDIR=/tmp
timeout 1500 python tools/parity_fullsize.py --configs 1,2,3,4,5 --weights ctor326,conditioned --seeds 3 --kinds shift,noise --fast \
    --out "$OUT/d_parity_batch.json" > "$OUT/d_parity_batch.txt" 2>&1
echo "rc=$?" >> "$OUT/d_parity_batch.txt"
grep -E "ALL|cases|rc=" "$OUT/d_parity_batch.txt" | cut -c1-250
timeout 300 python -m pytest tests/test_hip_parity_gpu.py -m gpu -x -q -k "survey_named or split_handoffs or rccl" 2>&1 | tail -3
