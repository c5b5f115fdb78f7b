#!/bin/bash
# Round 3, GPU call I: final evidence -- parity table at the BASELINE batch sizes, profiles (bench line, steady state, PMC passes), full -m gpu suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
timeout 900 python tools/parity_fullsize.py --configs 1,2,5,3,4 --weights ctor326,conditioned --seeds 3 --kinds shift,noise --fast \
    --cache gpurun_cache/parity --out "$OUT/i_parity_batch.json" > "$OUT/i_parity_batch.txt" 2>&1
echo "rc=$?" >> "$OUT/i_parity_batch.txt"
grep -E "ALL|cases|rc=" "$OUT/i_parity_batch.txt" | cut -c1-210
bash tools/collect_profiles.sh r03 > "$OUT/i_collect.log" 2>&1; tail -3 "$OUT/i_collect.log"
tail -c 400 "$OUT/r03_bench.json"
timeout 1100 python -m pytest tests -m gpu -x -q > "$OUT/i_pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/i_pytest.log"; tail -4 "$OUT/i_pytest.log"
