#!/bin/bash
# round-2 GPU call E: gsv4 (one wave per SIMD, 64 queries per wave) -- tests, A/B against gsv3, PMC
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "global_matching or propagation or scale_sweep or global_corr or gsv" 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -25 > "$OUT/e_gsv_tests.log"
{ for prec in exact fast; do for v3 in 0 1; do echo "== precision=$prec UM_GSV_V3=$v3"; UM_GSV_V3=$v3 timeout 120 python tools/bench_ops.py gsv --precision $prec --iters 20 2>&1 | grep -v "Warn\|amdgpu.ids"; done; done; } > "$OUT/e_ops_bench.log" 2>&1
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for prec in exact fast; do
    D=/tmp/pmc_e_${prec}
    (cd /tmp && timeout 200 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $D -o p -- python "$R/tools/bench_ops.py" gsv --precision $prec --iters 5 > "$OUT/e_pmc_${prec}.log" 2>&1 < /dev/null)
    F=$(find $D -name '*counter_collection.csv' | head -1)
    [ -n "$F" ] && python tools/pmc_summary.py "$F" gsv > "$OUT/e_pmc_gsv4_${prec}.json"
done
tail -8 "$OUT/e_gsv_tests.log"; cat "$OUT/e_ops_bench.log"
