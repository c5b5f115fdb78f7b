"""The LAST forward of a `rocprofv3 --kernel-trace --output-format csv` run, kernel by kernel in launch order (name, grid, duration):
which launch of a shared kernel (conv_kernel serves 1x1, 5x1, ... shapes) costs what.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python tools/profile_config.py gmflow_s2_rr6 4 512 768
    python tools/trace_sequence.py /tmp/tr/*/t_kernel_trace.csv [first-kernel-substring]
"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mark = sys.argv[2] if len(sys.argv) > 2 else 'pack7_kernel'
starts = [i for i, r in enumerate(rows) if mark in r['Kernel_Name']]
sel = rows[starts[-1]:]
t0 = int(sel[0]['Start_Timestamp'])
for r in sel:
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:60]
    grid = r.get('Grid_Size_X', r.get('Grid_Size', '?'))
    wg = r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  grid {grid:>9s} wg {wg:>4s}  {name}")
