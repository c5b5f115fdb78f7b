"""Per-kernel time of the LAST steps of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline
    python tools/steady_state.py gpurun_out/trace/t_kernel_trace.csv 10 > profiles/rNN_steady_state_breakdown.txt

A step starts at the stem's `pack7_kernel` launch (one per forward); the last N steps before the census forward / box probes that
end a bench.py run are summed per kernel name.
"""
import collections
import csv
import sys


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # round 6: bench.py ends with a key-tile census forward and the box probes (probe_*_kernel): the steady state ends before them
    cut = next((i for i, r in enumerate(rows) if 'probe_' in r['Kernel_Name']), len(rows))
    rows = rows[:cut]
    starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('void pack7_kernel') or r['Kernel_Name'].startswith('pack7_kernel')]
    if len(starts) < steps + 1:
        sys.exit(f'only {len(starts)} steps in the trace')
    sel = rows[starts[-steps - 1]:starts[-1]]            # the last step before the probes is the census forward: left out
    wall = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in sel:
        a = agg[r['Kernel_Name']]
        a[0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        a[1] += 1
    tot = sum(v[0] for v in agg.values())
    print(f'# last {steps} steps of the trace: kernel-sum {tot / steps:.3f} ms/step, wall {wall / steps:.3f} ms/step, '
          f'{len(sel) / steps:.0f} kernels/step')
    for name, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f'{ms / steps:8.3f} ms/step  {n / steps:6.1f} calls/step  {1e3 * ms / n:8.1f} us avg  {name[:120]}')


if __name__ == '__main__':
    main()
