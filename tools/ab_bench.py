"""Same-box A/B of whole-model throughput (GPU box).

MI355X boxes differ by +-5 % in `bench.py` throughput and a kernel's microbenchmark time is not its time inside the model
(a kernel launched back to back runs hotter and therefore at a lower clock than the same kernel between memory-bound
neighbours), so every optimisation of round 1 was accepted or dropped on an ABAB run of `bench.py` itself on ONE box:

    python tools/ab_bench.py --steps 30  A=UM_NO_MERGE=1  B=
    python tools/ab_bench.py  old=UM_LIB=unimatch_amd/_variants/libold.so  new=

Every argument is `label=ENV1=v1,ENV2=v2` (empty = the tree as it is).  Useful switches: `UM_LIB` (an alternative build of
the library, e.g. compiled with a -D flag into unimatch_amd/_variants/), `UM_NO_MERGE=1` (merge + LayerNorm as its own
launch), `UM_CONV_NO_ROWS=1` (generic convolution kernel only), `UM_CONV_NO_XCD=1` (plain workgroup order in the convolutions), `UM_CONV_PATCH=0` (no 2-D patch kernel; digits = tile widths it may serve),
`UM_SHORTCUT_F32=1` (encoder keeps fp32 copies for the identity shortcuts).  Run-to-run repeatability on one box is ~0.1 %.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    steps = '30'
    if '--steps' in args:
        i = args.index('--steps')
        steps = args[i + 1]
        del args[i:i + 2]
    variants = []
    for a in args:
        label, _, envs = a.partition('=')
        env = dict(kv.split('=', 1) for kv in envs.split(',') if kv)
        variants.append((label, env))
    if len(variants) < 2:
        sys.exit(__doc__)
    for rep in range(2):
        for label, extra in variants:
            env = dict(os.environ)
            env.update(extra)
            out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--no-fast', '--steps', steps],
                                 capture_output=True, text=True, env=env, cwd=ROOT)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                print(f'{label:20s} {d["value"]:9.2f} pairs/s  {d["ms_per_step"]:8.3f} ms/step  '
                      f'window_attn {d["roofline"]["avg_launch_ms"]:.4f} ms  gsv4 {d["roofline_global_corr"]["avg_launch_ms"]:.4f} ms  '
                      f'median {d.get("ms_per_step_median", 0):.3f}', flush=True)
            except (IndexError, ValueError, KeyError):
                print(f'{label:20s} FAILED\n{out.stderr[-800:]}', flush=True)


if __name__ == '__main__':
    main()
