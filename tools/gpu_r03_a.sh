#!/bin/bash
# Round 3, GPU call A: the whole -m gpu suite at HEAD, bench line, same-box A/B of the fma_mix hi|lo split, and the
# precision-budget experiments (P in one plane, gelu(H) in one plane) with their parity tables on config 2 at batch 8.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd "$R"
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -x -q > "$OUT/a_pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/a_pytest.log"
tail -5 "$OUT/a_pytest.log"
timeout 300 python bench.py > "$OUT/a_bench.json" 2> "$OUT/a_bench.err"; tail -c 600 "$OUT/a_bench.json"
V=unimatch_amd/_variants
timeout 400 python tools/ab_bench.py --steps 30 pre_mix=UM_LIB=$V/libpre_mix.so new= p1=UM_LIB=$V/libp1.so ffnh1=UM_LIB=$V/libffnh1.so > "$OUT/a_ab.txt" 2>&1
cat "$OUT/a_ab.txt"
C=/tmp/um_parity_cache
for lib in main p1 ffnh1; do
  if [ $lib = main ]; then L=""; else L="$V/lib$lib.so"; fi
  UM_LIB=$L timeout 900 python tools/parity_fullsize.py --configs 2 --weights ctor326,conditioned --seeds 3 --cache $C \
      --out "$OUT/a_parity_cfg2_$lib.json" > "$OUT/a_parity_cfg2_$lib.txt" 2>&1
  echo "== $lib"; grep -E "ALL|cases" "$OUT/a_parity_cfg2_$lib.txt"
done
