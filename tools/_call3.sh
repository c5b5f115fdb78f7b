#!/bin/bash
# scratch driver of one gpurun call (round 4, call 3)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=gpurun_out; mkdir -p $O
V=unimatch_amd/_variants
( python tools/bench_layer_kernels.py 30
  for n in ffn_p1 ffn_p2 ffn_p3 ffn_p4 ffn_p5 attn_off32; do UM_LIB=$V/lib$n.so python tools/bench_layer_kernels.py 30; done
  UM_LIB=$V/libdbg.so UM_WATTN_FORCE_SPLIT=2 python tools/bench_layer_kernels.py 30 | sed 's/libdbg.so   /libdbg+split2/'
  python tools/bench_layer_kernels.py 30 ) 2>&1 | grep -v amdgpu.ids > $O/c4_layer_kernels.txt
cat $O/c4_layer_kernels.txt
UM_LIB=$V/libffntrace_p2.so python tools/trace_ffn.py 2>&1 | grep -v amdgpu.ids > $O/c4_ffn_trace_p2.txt
BEST=$(python - <<'PY'
import re
best, name = 1e9, 'ffn_p2'
for line in open('gpurun_out/c4_layer_kernels.txt'):
    m = re.match(r'lib(ffn_p\d)\.so.*ffn ([0-9.]+) ms', line)
    if m and float(m.group(2)) < best:
        best, name = float(m.group(2)), m.group(1)
print(name)
PY
)
echo "best ffn variant: $BEST" | tee $O/c4_best.txt
python tools/ab_bench.py --steps 30 head= $BEST=UM_LIB=$V/lib$BEST.so split2=UM_LIB=$V/libdbg.so,UM_WATTN_FORCE_SPLIT=2 off32=UM_LIB=$V/libattn_off32.so 2>&1 | grep -v amdgpu.ids > $O/c4_ab.txt
cat $O/c4_ab.txt
bash tools/collect_gather_profiles.sh r04 > $O/c4_collect.log 2>&1
tail -5 $O/c4_collect.log
