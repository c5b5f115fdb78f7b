#!/bin/bash
# Round-4 evidence for the gather / short-reduction kernels (K3, K4, K6, K7 against HBM and fp32 VALU) and for
# configs 3 / 4 / 5 as a whole, in ONE gpurun call (run from the repo root on the GPU box):
#   gpurun --timeout 900 -- 'bash tools/collect_gather_profiles.sh r04'
# per config: rocprofv3 --kernel-trace --stats (per-kernel time), one --pmc pass with FETCH_SIZE + SQ + GRBM counters (different
# counter blocks: they fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") and one --pmc pass with WRITE_SIZE (TCC slots).
# Then tools/bench_ops.py local (hipEvent durations against the compulsory bytes / algorithmic FLOPs of SURVEY 8(d)).
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
GATHER="k4m_kernel local_corr_with_flow local_corr_softmax prop_local_attn depth_corr_softmax feat_planes split_planes"
run_cfg() {   # name batch H W label
  local name=$1 b=$2 hh=$3 ww=$4 lab=$5
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/${TAG}_${lab}_t" -o p -- \
      python "$R/tools/profile_config.py" $name $b $hh $ww > "$OUT/${TAG}_${lab}_trace.log" 2>&1 < /dev/null)
  local ST=$(find "/tmp/${TAG}_${lab}_t" -name '*kernel_stats.csv' | head -1)
  [ -n "$ST" ] && cp "$ST" "$OUT/${TAG}_${lab}_kernel_stats.csv"
  (cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d "/tmp/${TAG}_${lab}_f" -o p -- \
      python "$R/tools/profile_config.py" $name $b $hh $ww > "$OUT/${TAG}_${lab}_pmcf.log" 2>&1 < /dev/null)
  local PMC=$(find "/tmp/${TAG}_${lab}_f" -name '*counter_collection.csv' | head -1)
  [ -n "$PMC" ] && python tools/pmc_summary.py "$PMC" $GATHER window_attn ffn_kernel gsv > "$OUT/${TAG}_${lab}_pmc_fetch_sq.json"
  (cd /tmp && timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "/tmp/${TAG}_${lab}_w" -o p -- \
      python "$R/tools/profile_config.py" $name $b $hh $ww > "$OUT/${TAG}_${lab}_pmcw.log" 2>&1 < /dev/null)
  PMC=$(find "/tmp/${TAG}_${lab}_w" -name '*counter_collection.csv' | head -1)
  [ -n "$PMC" ] && python tools/pmc_summary.py "$PMC" $GATHER window_attn ffn_kernel gsv > "$OUT/${TAG}_${lab}_pmc_write.json"
  rm -rf "/tmp/${TAG}_${lab}_t" "/tmp/${TAG}_${lab}_f" "/tmp/${TAG}_${lab}_w"
}
run_cfg gmstereo_s2_rr3 4 512 960 cfg3
run_cfg gmflow_s2_rr6 4 512 768 cfg4
run_cfg gmdepth_s1 16 480 640 cfg5
timeout 200 python tools/bench_ops.py local --iters 20 > "$OUT/${TAG}_ops_local.txt" 2>&1
ls -la "$OUT" | grep "${TAG}_" | tail -30
