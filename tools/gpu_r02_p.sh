#!/bin/bash
# round-2 GPU call P: full GPU suite + bench at the window_attn micro-optimisation state
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -8 > "$OUT/p_gpu_tests.log"
timeout 400 python bench.py > "$OUT/p_bench.json" 2> "$OUT/p_bench.err"
timeout 200 python tools/bench_configs.py --steps 10 2>&1 | grep cfg > "$OUT/p_configs.log"
tail -3 "$OUT/p_gpu_tests.log"; cat "$OUT/p_configs.log"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/p_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline_global_corr']['frac'], d['roofline_global_corr']['avg_launch_ms'], d['fast']['value'], d['epe'])
PY
