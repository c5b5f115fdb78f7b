#!/bin/bash
# round-2 GPU call W: query-projection geometry sweep + kernel statistics of config 5 (depth) and config 3 (stereo)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity_gpu.py -q -x -k "query_projection or attention_merge" 2>&1 | grep -v "Warn\|amdgpu.ids" | grep -v "^$" | tail -25 > "$OUT/w_tests.log"
for spec in "cfg5 gmdepth_s1 16 480 640" "cfg3 gmstereo_s2_rr3 4 512 960"; do
  set -- $spec
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/w_$1" -o p -- \
      python "$R/tools/profile_config.py" $2 $3 $4 $5 > "$OUT/w_$1.log" 2>&1 < /dev/null)
  rm -f "$OUT/w_$1"/*kernel_trace.csv
done
tail -3 "$OUT/w_tests.log"; find "$OUT/w_cfg5" -name '*kernel_stats.csv' | head -1 | xargs head -8 | cut -c1-160
