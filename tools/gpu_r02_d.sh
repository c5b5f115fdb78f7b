#!/bin/bash
# round-2 GPU call D: PMC counters of gsv3 (exact, fast) and the round-1 kernel (exact) -- where do the wave cycles go?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
for cfg in "0 exact" "0 fast" "1 exact"; do
  set -- $cfg; v2=$1; prec=$2
  for p in 1 2; do
    if [ $p = 1 ]; then PMC="$P1"; else PMC="$P2"; fi
    D=/tmp/pmc_${v2}_${prec}_$p
    (cd /tmp && UM_GSV_V2=$v2 timeout 200 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $D -o p -- python "$R/tools/bench_ops.py" gsv --precision $prec --iters 5 > "$OUT/d_pmc_${v2}_${prec}_$p.log" 2>&1 < /dev/null)
    F=$(find $D -name '*counter_collection.csv' | head -1)
    [ -n "$F" ] && python tools/pmc_summary.py "$F" gsv > "$OUT/d_pmc_gsv_v2${v2}_${prec}_$p.json"
  done
done
ls -la $OUT | tail -20
