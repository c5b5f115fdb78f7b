#!/bin/bash
# Device ISA + resource usage of one csrc file:  tools/isa.sh window_attn [kernel-name-substring]  -> /tmp/<file>.s (+ /tmp/<file>_k.s)
# The build's flags (unimatch_amd/build.py); prints VGPRs / scratch of every kernel whose mangled name contains the substring.
F=$1; K=${2:-}
cd /root/repo/unimatch_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize ${ISA_FLAGS:-} -S --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage -o /tmp/$F.s $F.hip 2>&1 | grep -v "hip-link" | grep -E "error|Function Name|VGPRs:|ScratchSize" -A0 | \
    sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - 2>/dev/null | grep -E "error|$K"
