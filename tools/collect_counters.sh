#!/bin/bash
# LDS / VALU / VMEM / L2 counter passes over bench.py for the kernels that bound the hot path (window_attn_kernel, ffn_kernel,
# gsv4_kernel) -- VERDICT r04 "next round" items 1(i) and 2.  One gpurun call, ~4 GPU-minutes:
#   gpurun --timeout 900 -- 'bash tools/collect_counters.sh r05'
# Every pass is its own rocprofv3 run with --kernel-trace only (the pool refuses --pmc together with the tracing domains); SQ has
# 8 counter slots per pass, TCC 4, GRBM 2 (MI355X_MICROARCH.md "rocprofv3 PMC slots").  Output: gpurun_out/<tag>_pmc_<pass>.json
# (tools/pmc_summary.py: mean per dispatch and kernel), merged by tools/pmc_bounds.py into profiles/.
set -u
TAG=${1:-rNN}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
KERNELS="${KERNELS:-window_attn ffn_kernel gsv4_kernel kv4_kernel}"
pass() {   # pass <name> <counters...>
    local name=$1; shift
    (cd /tmp && timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/${TAG}_$name -o p -- \
        python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-fast --streams 1 > "$OUT/${TAG}_pmc_$name.log" 2>&1 < /dev/null)
    local f
    f=$(find /tmp/${TAG}_$name -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python tools/pmc_summary.py "$f" $KERNELS > "$OUT/${TAG}_pmc_$name.json"; else echo "pass $name: no counter file"; tail -5 "$OUT/${TAG}_pmc_$name.log"; fi
    rm -rf /tmp/${TAG}_$name
}
pass lds1 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass lds2 SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass sq3 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU GRBM_GUI_ACTIVE
pass tcc1 TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_READ_sum
pass tcc3 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
ls -la "$OUT" | tail -20
