#!/bin/bash
# round-2 GPU call F: timing ablations of gsv4 (exact): where do the non-MFMA cycles go?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
{ for rep in 1 2; do for abl in 0 1 2 3 4 5; do echo "== UM_GSV4_ABL=$abl"; UM_GSV4_ABL=$abl timeout 120 python tools/bench_ops.py gsv --precision exact --iters 20 2>&1 | grep "corr flow"; done; done
  echo "== zeros"; timeout 120 python tools/bench_ops.py gsv --precision exact --iters 20 --zeros 2>&1 | grep "corr flow"; } > "$OUT/f_ablation.log" 2>&1
cat "$OUT/f_ablation.log"
