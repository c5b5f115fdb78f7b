#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out; mkdir -p $O
bash tools/collect_profiles.sh r04 > $O/j4_collect.log 2>&1
tail -3 $O/j4_collect.log
timeout 300 python bench.py --workload cfg4 --steps 5 --warmup 2 --no-fast > $O/r04_bench_cfg4.json 2> $O/r04_bench_cfg4.err
UM_BENCH_FORCE_DIST=1 timeout 300 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-fast --no-cpu-baseline > $O/r04_bench_cfg4_dist1.json 2> $O/r04_bench_cfg4_dist1.err
timeout 200 python tools/bench_modelzoo.py > $O/r04_modelzoo.txt 2>&1
python -m pytest tests -q -m gpu > $O/j4_gpu_tests.log 2>&1
grep -v amdgpu.ids $O/j4_gpu_tests.log | tail -6
