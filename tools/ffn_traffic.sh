#!/bin/bash
# Fabric read requests of ffn_kernel with and without its residual read (where do 183 MB of reads for 100.7 MB of inputs come
# from?).  One gpurun call:  bash tools/ffn_traffic.sh   (needs unimatch_amd/_variants/libffnnores.so = -DUM_FFN_ABL=32: residual not read)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; mkdir -p gpurun_out
for v in ${FT_VARIANTS:-shipped ffnnores}; do
  L=""; [ $v != shipped ] && L="$R/unimatch_amd/_variants/lib$v.so"
  (cd /tmp && UM_LIB=$L timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/ft_$v -o p -- \
      python "$R/tools/bench_layer_kernels.py" 5 > "$R/gpurun_out/ffn_traffic_$v.log" 2>&1 < /dev/null)
  f=$(find /tmp/ft_$v -name '*counter_collection.csv' | head -1)
  echo "== $v"; python tools/pmc_summary.py "$f" ffn_kernel
done
