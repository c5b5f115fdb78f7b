#!/bin/bash
# round-2 final GPU call: full GPU suite, bench line, FETCH_SIZE / WRITE_SIZE passes (pmc_current.json), full-size parity table
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp; TAG=r02f
timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -3 > "$OUT/${TAG}_tests.log"
timeout 240 python bench.py 2> "$OUT/${TAG}_bench.err" | tail -1 > "$OUT/${TAG}_bench.json"
(cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/${TAG}_pmc -o p -- \
    python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-fast > "$OUT/${TAG}_pmc.log" 2>&1 < /dev/null)
PMC=$(find /tmp/${TAG}_pmc -name '*counter_collection.csv' | head -1)
if [ -n "$PMC" ]; then python tools/pmc_summary.py "$PMC" conv_ nhwc_apply window_attn gsv ffn_kernel linear_kernel > "$OUT/${TAG}_pmc_fetch.json"; fi
(cd /tmp && timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/${TAG}_pmcw -o p -- \
    python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-fast > "$OUT/${TAG}_pmcw.log" 2>&1 < /dev/null)
PMC=$(find /tmp/${TAG}_pmcw -name '*counter_collection.csv' | head -1)
if [ -n "$PMC" ]; then python tools/pmc_summary.py "$PMC" window_attn gsv > "$OUT/${TAG}_pmc_write.json"; fi
timeout 400 python tools/parity_fullsize.py --fast --out "$OUT/${TAG}_parity.json" > "$OUT/${TAG}_parity.txt" 2>&1
cat "$OUT/${TAG}_tests.log"; tail -11 "$OUT/${TAG}_parity.txt"
