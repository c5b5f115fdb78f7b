#!/bin/bash
# Fabric reads of the cost-volume launches of config 4 (random-init weights: incoherent flow), target-ordered against natural tiles:
#   gpurun -- 'bash tools/k4_traffic.sh'    (the pixel path fetched 20 x the compulsory bytes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$R"; mkdir -p gpurun_out
for f in 0 2; do
  (cd /tmp && timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d /tmp/k4t_$f -o p -- \
      python "$R/tools/bench_configs.py" --only 4 --steps 2 --k4-flags $f > "$R/gpurun_out/k4_traffic_$f.log" 2>&1 < /dev/null)
  echo "== k4 flags $f (0: target-ordered where the flow is incoherent, 2: natural tiles only)"
  python tools/pmc_summary.py "$(find /tmp/k4t_$f -name '*counter_collection.csv' | head -1)" k4m_kernel k4s_
done
