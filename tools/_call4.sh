#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out; mkdir -p $O
python -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "kv4 or ffn or fused_layer or transformer or e2e or end_to_end or golden or graph" 2>&1 | grep -v amdgpu.ids | tail -25 > $O/d4_tests.log
cat $O/d4_tests.log | tail -8
python tools/ab_bench.py --steps 30 head= kvsolo=--set,HipOps.fused_kv=0 kv2=--set,HipOps.block_kv=0 2>&1 | grep -v amdgpu.ids > $O/d4_ab.txt
cat $O/d4_ab.txt
python bench.py --no-cpu-baseline --no-fast --steps 20 2>/dev/null > $O/d4_bench.json
python -c "
import json; d=json.loads(open('$O/d4_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['hot_path_kernels_ms_per_step'])"
