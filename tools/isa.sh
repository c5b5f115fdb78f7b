#!/bin/bash
# Device ISA + resource usage of one csrc file:  tools/isa.sh window_attn [kernel-name-substring]  -> /tmp/<file>.s
# The build's flags (unimatch_amd/build.py: FLAGS + EXTRA_FLAGS of the file); prints VGPRs / scratch of every kernel whose mangled name
# contains the substring.
F=$1; K=${2:-}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
flags=$(cd "$ROOT" && python -c "from unimatch_amd.build import FLAGS, EXTRA_FLAGS; print(' '.join(FLAGS + EXTRA_FLAGS.get('$F.hip', [])))")
cd "$ROOT/unimatch_amd/csrc" || exit 1
/opt/rocm/bin/hipcc $flags ${ISA_FLAGS:-} -S --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage -o /tmp/$F.s $F.hip 2>&1 | grep -v "hip-link" | grep -E "error|Function Name|VGPRs:|ScratchSize" -A0 | \
    sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - 2>/dev/null | grep -E "error|$K"
