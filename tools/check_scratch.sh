#!/bin/bash
# every kernel instantiation of the shipped library must have ScratchSize 0 (no register spill).
#   tools/check_scratch.sh            -> lists every kernel with a non-zero ScratchSize and exits non-zero if there is one
# The flags are the build's own (unimatch_amd/build.py: FLAGS + EXTRA_FLAGS per file).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/unimatch_amd/csrc" || exit 2
bad=0
for f in *.hip; do
  [ "$f" = microbench.hip ] && continue
  flags=$(cd "$ROOT" && python -c "from unimatch_amd.build import FLAGS, EXTRA_FLAGS; print(' '.join(FLAGS + EXTRA_FLAGS.get('$f', [])))")
  out=$(/opt/rocm/bin/hipcc $flags --cuda-device-only -c "$f" -o /dev/null \
        -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - -)
  n=$(echo "$out" | grep -c "Function Name")
  spill=$(echo "$out" | grep -v "ScratchSize \[bytes/lane\]: 0$")
  echo "$f: $n kernels"
  if [ -n "$spill" ]; then echo "$spill" | sed 's/^/    SPILL: /'; bad=1; fi
done
exit $bad
