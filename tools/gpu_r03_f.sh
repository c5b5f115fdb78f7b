#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p "$OUT"; cd "$R"
V=unimatch_amd/_variants
echo "== trace pipe"; UM_LIB=$V/libtracepipe.so timeout 120 python tools/trace_attn.py A,B,C,wait,barrier 2>&1 | grep "^wg" | tee "$OUT/f_trace_pipe.txt"
echo "== trace head"; UM_LIB=$V/libtrace.so timeout 120 python tools/trace_attn.py prep,QK,bias,softmaxPV,wait,barrier 2>&1 | grep "^wg" | tee "$OUT/f_trace_head.txt"
