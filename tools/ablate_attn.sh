#!/bin/bash
# Build ablated variants of the attention kernel (diagnostics) into gpurun-shippable libs: tools/abl/lib_<bits>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/abl
for bits in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DUM_ABL=$bits -shared -o tools/abl/lib_$bits.so \
    unimatch_amd/csrc/capi.hip unimatch_amd/csrc/global_match.hip unimatch_amd/csrc/window_attn.hip unimatch_amd/csrc/local_ops.hip
done
