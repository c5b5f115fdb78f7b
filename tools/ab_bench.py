"""Same-box A/B of whole-model throughput (GPU box).

MI355X boxes differ by +-5 % in `bench.py` throughput and a kernel's microbenchmark time is not its time inside the model
(a kernel launched back to back runs hotter and therefore at a lower clock than the same kernel between memory-bound
neighbours), so every optimisation was accepted or dropped on an ABAB run of `bench.py` itself on ONE box:

    python tools/ab_bench.py  old=UM_LIB=unimatch_amd/_variants/libold.so  new=
    python tools/ab_bench.py --steps 30  A=--set,HipOps.fused_merge=0  B=

Every argument is `label=SPEC1,SPEC2,...` (empty = the tree as it is).  A SPEC is either `ENV=value` -- `UM_LIB=<path>` loads a
diagnostic build of the library (`python -m unimatch_amd.build --variant NAME -DFLAG ...` -> unimatch_amd/_variants/libNAME.so;
with `-DUM_DEBUG_SWITCHES` the library's own A/B switches `UM_GSV_V3`, `UM_CONV_PATCH`, `UM_CONV_NO_ROWS`, `UM_CONV_NO_XCD`,
`UM_WATTN_NO_KSPLIT`, `UM_FFN_NO_HSPLIT` become live; the shipped library reads no environment variable) -- or a `bench.py`
argument starting with `--` (`--set,HipOps.fused_merge=0`: merge + LayerNorm as its own launch; `--set,HipOps.fused_qproj=0`;
`--set,CNNEncoder.shortcut_f32=1`).  Run-to-run repeatability on one box is ~0.1 %.
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
        label, _, specs = a.partition('=')
        parts = [p for p in specs.split(',') if p]
        env, extra, i = {}, [], 0
        while i < len(parts):
            if parts[i] == '--set' and i + 1 < len(parts):          # bench.py --set Class.attr=value
                extra += parts[i:i + 2]
                i += 2
            elif parts[i].startswith('--'):                          # any other bench.py flag
                extra.append(parts[i])
                i += 1
            else:                                                    # ENV=value
                k, _, v = parts[i].partition('=')
                env[k] = v
                i += 1
        variants.append((label, (env, extra)))
    if len(variants) < 2:
        sys.exit(__doc__)
    for rep in range(2):
        for label, (extra_env, extra_args) in variants:
            env = dict(os.environ)
            env.update(extra_env)
            out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--no-fast', '--steps', steps] + extra_args,
                                 capture_output=True, text=True, env=env, cwd=ROOT)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                print(f'{label:20s} {d["value"]:9.2f} pairs/s  {d["ms_per_step"]:8.3f} ms/step  '
                      f'window_attn {d["roofline"]["avg_launch_ms"]:.4f} ms  gsv4 {d["roofline_global_corr"]["avg_launch_ms"]:.4f} ms  '
                      f'ffn {(d.get("roofline_ffn") or {}).get("avg_launch_ms", 0):.4f} ms  hot path {d.get("hot_path_ms_per_step", 0):.3f} ms  '
                      f'encoder {d.get("encoder_ms_per_step", 0):.3f} ms  median {d.get("ms_per_step_median", 0):.3f}', flush=True)
            except (IndexError, ValueError, KeyError):
                print(f'{label:20s} FAILED\n{out.stderr[-800:]}', flush=True)


if __name__ == '__main__':
    main()
