#!/bin/bash
# Round 3, GPU call L: final state -- full -m gpu suite, batch-1 latency A/B of the ticket hand-off, bench line, PMC passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -x -q > "$OUT/l_pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/l_pytest.log"; grep -E "passed|failed|rc=" "$OUT/l_pytest.log" | tail -3
for rep in 1 2; do for L in head0 new; do
  if [ $L = new ]; then LIB=""; else LIB=unimatch_amd/_variants/lib$L.so; fi
  UM_LIB=$LIB timeout 120 python bench.py --batch 1 --steps 100 --warmup 10 --no-cpu-baseline --no-fast 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L batch1', d['ms_per_step'], 'ms/step', d['ms_per_step_median'], 'median')"
done; done | tee "$OUT/l_batch1_ab.txt"
timeout 300 python bench.py > "$OUT/r03_bench_final.json" 2> "$OUT/r03_bench_final.err"; tail -c 300 "$OUT/r03_bench_final.json"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  T=$(echo $C | cut -c1-5)
  (cd /tmp && timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/l_pmc_$T -o p -- \
      python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-fast > "$OUT/l_pmc_$T.log" 2>&1 < /dev/null)
  P=$(find /tmp/l_pmc_$T -name '*counter_collection.csv' | head -1)
  if [ -n "$P" ]; then python tools/pmc_summary.py "$P" window_attn gsv ffn_kernel > "$OUT/l_pmc_$T.json"; fi
done
ls -la "$OUT" | grep "l_pmc"
