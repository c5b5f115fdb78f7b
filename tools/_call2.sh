#!/bin/bash
# scratch driver of one gpurun call (round 4, call 2)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out; mkdir -p $O
V=unimatch_amd/_variants
# 1. screening of the layer kernels
( python tools/bench_layer_kernels.py 30
  for n in ffn0 attn_dma1 attn_off32 attn_both; do UM_LIB=$V/lib$n.so python tools/bench_layer_kernels.py 30; done
  python tools/bench_layer_kernels.py 30 ) > $O/b4_layer_kernels.txt 2>&1
# 2. FFN section stamps, new order and old
UM_LIB=$V/libffntrace.so python tools/trace_ffn.py > $O/b4_ffn_trace_order1.txt 2>&1
UM_LIB=$V/libffntrace0.so python tools/trace_ffn.py > $O/b4_ffn_trace_order0.txt 2>&1
# 3. whole-model ABAB: head (ffn order 1, one k|v launch per block) / old FFN order / two k|v launches per block
python tools/ab_bench.py --steps 30 head= ffn0=UM_LIB=$V/libffn0.so kv2=--set,HipOps.block_kv=0 attn_dma1=UM_LIB=$V/libattn_dma1.so > $O/b4_ab.txt 2>&1
# 4. CPU legs in the background from here on (timing-sensitive parts are done)
(python tools/stage_parity.py --stage cpu --configs 4 --kinds shift --cache /tmp/stage --workers 4 --threads 16 > $O/b4_stage_cpu.log 2>&1 &)
(python tools/parity_fullsize.py --stage cpu --configs 2 --weights ctor326 --kinds shift --seeds 1 --cache /tmp/pf --workers 8 --threads 8 > $O/b4_pf_cpu.log 2>&1 &)
python -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "ffn or fused_layer or linear or transformer" > $O/b4_tests.log 2>&1
grep -E "passed|failed" $O/b4_tests.log
# 5. precision budget of the convolutions: config 2 end to end (encoder + upsampler head), config 4 stage by stage (refinement block)
for n in head conv2p1 conv2p2; do
  L=""; [ $n != head ] && L=$V/lib$n.so
  UM_LIB=$L python tools/parity_fullsize.py --configs 2 --weights ctor326 --kinds shift --seeds 1 --cache /tmp/pf --workers 8 --threads 8 > $O/b4_budget_cfg2_$n.txt 2>&1
  UM_LIB=$L python tools/stage_parity.py --configs 4 --kinds shift --blocks 0 --cache /tmp/stage --workers 4 --threads 16 > $O/b4_budget_cfg4_$n.txt 2>&1
  UM_LIB=$L python bench.py --no-cpu-baseline --no-fast --steps 20 > $O/b4_bench_$n.json 2> /dev/null
done
ls -la $O | grep b4_
