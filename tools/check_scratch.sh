#!/bin/bash
# every kernel instantiation of the shipped library must have ScratchSize 0 (no register spill).
#   tools/check_scratch.sh            -> lists every kernel with a non-zero ScratchSize and exits non-zero if there is one
cd /root/repo/unimatch_amd/csrc 2>/dev/null || cd "$(dirname "$0")/../unimatch_amd/csrc" || exit 2
bad=0
for f in *.hip; do
  [ "$f" = microbench.hip ] && continue
  extra=""; case $f in ffn.hip|global_match.hip|window_attn.hip|linear.hip) extra="-fno-slp-vectorize";; esac
  out=$(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result $extra --cuda-device-only -c "$f" -o /dev/null \
        -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - -)
  n=$(echo "$out" | grep -c "Function Name")
  spill=$(echo "$out" | grep -v "ScratchSize \[bytes/lane\]: 0$")
  echo "$f: $n kernels"
  if [ -n "$spill" ]; then echo "$spill" | sed 's/^/    SPILL: /'; bad=1; fi
done
exit $bad
