#!/bin/bash
# round-2 GPU call G: PMC of gsv4 ablations (0 = real, 2 = no softmax fillers) and zero operands: busy fraction and effective clock
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for cfg in "0 x" "2 x" "0 --zeros" "5 x"; do
    set -- $cfg; abl=$1; z=$2; [ "$z" = x ] && z=""
    D=/tmp/pmc_g_${abl}_${z#--}
    (cd /tmp && UM_GSV4_ABL=$abl timeout 200 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $D -o p -- python "$R/tools/bench_ops.py" gsv --precision exact --iters 5 $z > "$OUT/g_pmc_${abl}_${z#--}.log" 2>&1 < /dev/null)
    F=$(find $D -name '*counter_collection.csv' | head -1)
    [ -n "$F" ] && python tools/pmc_summary.py "$F" gsv > "$OUT/g_pmc_abl${abl}_${z#--}.json"
    K=$(find $D -name '*kernel_trace.csv' | head -1)
    [ -n "$K" ] && python - "$K" <<'PY' > "$OUT/g_dur_abl${abl}_${z#--}.txt"
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    if 'gsv' in k: print(k, 'n',len(v),'mean_us %.1f min_us %.1f'%(sum(v)/len(v),min(v)))
PY
done
for f in $OUT/g_dur_*.txt; do echo $f; cat $f; done
