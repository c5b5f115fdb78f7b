#!/bin/bash
# Evidence for profiles/ in ONE gpurun call (run from the repo root on the GPU box; ~3 GPU-minutes):
#   gpurun --timeout 600 -- 'bash tools/collect_profiles.sh r02'
# writes gpurun_out/<tag>_*: the bench.py line, the steady-state per-kernel breakdown of a traced bench.py, a FETCH_SIZE pass,
# all five BASELINE configs, and the config-4 kernel statistics.  Copy what should be judged into profiles/.
# The traced / counted runs pass --streams 1 (one forward per step, launches serialised: the mode bench.py's roofline durations are
# measured in; per-kernel durations of concurrent forwards overlap and are not per-kernel figures); the bench line itself is the default.
# Pitfalls this script encodes: rocprofv3 needs TMPDIR=/tmp and an explicit --output-format csv (the default is a database);
# its stdin must not be the terminal; --pmc goes with --kernel-trace only; the raw kernel trace is large -- reduce it on the box.
set -u
TAG=${1:-rNN}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
timeout 240 python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -o t -- \
    python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-fast --streams 1 > "$OUT/${TAG}_trace.log" 2>&1 < /dev/null)
TRACE=$(find "$OUT/${TAG}_trace" -name '*kernel_trace.csv' | head -1)
if [ -n "$TRACE" ]; then python tools/steady_state.py "$TRACE" 10 > "$OUT/${TAG}_steady_state.txt"; rm -f "$TRACE"; fi
# the same trace of the DEFAULT command (two concurrent forwards of half the batch): kernel statistics only -- the durations of kernels that
# share the GPU overlap, so their sum exceeds the wall time and they do not price a kernel; kept to show what the headline steps launch
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_default" -o t -- \
    python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-fast > "$OUT/${TAG}_trace_default.log" 2>&1 < /dev/null)
rm -f "$OUT/${TAG}_trace_default"/*kernel_trace.csv
(cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/${TAG}_pmc -o p -- \
    python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-fast --streams 1 > "$OUT/${TAG}_pmc.log" 2>&1 < /dev/null)
PMC=$(find /tmp/${TAG}_pmc -name '*counter_collection.csv' | head -1)
if [ -n "$PMC" ]; then python tools/pmc_summary.py "$PMC" conv_ nhwc_apply window_attn gsv ffn_kernel linear_kernel kv4_kernel > "$OUT/${TAG}_pmc_fetch.json"; fi
(cd /tmp && timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/${TAG}_pmcw -o p -- \
    python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-fast --streams 1 > "$OUT/${TAG}_pmcw.log" 2>&1 < /dev/null)
PMC=$(find /tmp/${TAG}_pmcw -name '*counter_collection.csv' | head -1)
if [ -n "$PMC" ]; then python tools/pmc_summary.py "$PMC" window_attn gsv ffn_kernel kv4_kernel > "$OUT/${TAG}_pmc_write.json"; fi
(cd /tmp && timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d /tmp/${TAG}_pmcs -o p -- \
    python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-fast --streams 1 > "$OUT/${TAG}_pmcs.log" 2>&1 < /dev/null)
PMC=$(find /tmp/${TAG}_pmcs -name '*counter_collection.csv' | head -1)
if [ -n "$PMC" ]; then python tools/pmc_summary.py "$PMC" window_attn gsv ffn_kernel > "$OUT/${TAG}_pmc_sq.json"; fi
(cd /tmp && timeout 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d /tmp/${TAG}_pmcl -o p -- \
    python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-fast --streams 1 > "$OUT/${TAG}_pmcl.log" 2>&1 < /dev/null)
PMC=$(find /tmp/${TAG}_pmcl -name '*counter_collection.csv' | head -1)
if [ -n "$PMC" ]; then python tools/pmc_summary.py "$PMC" window_attn gsv ffn_kernel > "$OUT/${TAG}_pmc_lds.json"; fi
# then, in the build container:  python tools/pmc_roofline.py gpurun_out/${TAG}_pmc_fetch.json gpurun_out/${TAG}_pmc_write.json gpurun_out/${TAG}_pmc_sq.json gpurun_out/${TAG}_pmc_lds.json
timeout 200 python tools/bench_configs.py --only 1,2,3,4,5,10 --steps 10 2>&1 | grep -E "cfg|\(10\)" > "$OUT/${TAG}_all_configs.txt"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_cfg4" -o p -- \
    python "$R/tools/profile_config.py" gmflow_s2_rr6 4 512 768 > "$OUT/${TAG}_cfg4.log" 2>&1 < /dev/null)
rm -f "$OUT/${TAG}_cfg4"/*kernel_trace.csv
ls -la "$OUT" | tail -12
