#!/bin/bash
# round-2 GPU call B: gsv3 after the NaN fix, PMC counters v3 vs v2, power experiment (zero operands), RCCL world-1 diagnosis, all GPU tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -k "global_matching or propagation or scale_sweep" 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -15 > "$OUT/b_gsv_tests.log"
{ for z in "" "--zeros"; do for v2 in 0 1; do echo "== exact UM_GSV_V2=$v2 $z"; UM_GSV_V2=$v2 timeout 120 python tools/bench_ops.py gsv --precision exact --iters 20 $z 2>&1 | grep -v "Warn\|amdgpu.ids"; done; done; } > "$OUT/b_power.log" 2>&1
for v2 in 0 1; do
  (cd /tmp && UM_GSV_V2=$v2 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_gsv_$v2 -o p -- python "$R/tools/bench_ops.py" gsv --precision exact --iters 5 > "$OUT/b_pmc_$v2.log" 2>&1 < /dev/null)
  F=$(find /tmp/pmc_gsv_$v2 -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && python tools/pmc_summary.py "$F" gsv > "$OUT/b_pmc_gsv_v2_$v2.json"
  (cd /tmp && UM_GSV_V2=$v2 timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc2_gsv_$v2 -o p -- python "$R/tools/bench_ops.py" gsv --precision exact --iters 5 > "$OUT/b_pmc2_$v2.log" 2>&1 < /dev/null)
  F=$(find /tmp/pmc2_gsv_$v2 -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && python tools/pmc_summary.py "$F" gsv > "$OUT/b_pmc2_gsv_v2_$v2.json"
done
UM_BENCH_FORCE_DIST=1 NCCL_DEBUG=WARN timeout 200 python -X faulthandler bench.py --no-cpu-baseline --no-fast --steps 5 --warmup 2 > "$OUT/b_bench_dist1.json" 2> "$OUT/b_bench_dist1.err"; echo "rc=$?" >> "$OUT/b_bench_dist1.err"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -30 > "$OUT/b_gpu_tests.log"
cat "$OUT/b_gsv_tests.log" | tail -3; cat "$OUT/b_power.log"; tail -5 "$OUT/b_bench_dist1.err"; tail -5 "$OUT/b_gpu_tests.log"
