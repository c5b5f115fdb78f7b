"""Stage-isolated (teacher-forced) parity of BASELINE configs 3 and 4 on the REFERENCE CONSTRUCTOR's seed-326 weights, at the
batch they are measured at (4 samples, 512x960 and 512x768), both image kinds: every stage of the product -- encoder (both
scales), Transformer at both scales (whole, and block by block, incl. the 1-D swin cross layers of the stereo model), global
matching, global propagation, x2 flow up-scaling + warp, local matching (2-D / 1-D), local propagation, every refinement
iteration's cost volume and K4 + update block, convex upsampling -- is fed the fp64 ORACLE's input of that stage and compared
with the oracle's fp64 output of that stage, next to the fp32 CPU port of the same stage on the same inputs.

Gate: per stage, EVERY sample: GPU error <= 2 x the port's error (+ 4 fp32 ulps of the stage's magnitude).
Reference: unimatch/unimatch.py:136-354, matching.py:39-123,154-200, attention.py:107-163,217-253 (tools/stage_parity.py lists
the file:line of every stage).  Table of the full run: profiles/r04_stage_parity.txt.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import stage_parity as sp  # noqa: E402

pytestmark = pytest.mark.gpu
SEED = 1000
CASES = [(cfg, kind) for cfg in (3, 4) for kind in ('shift', 'noise')]


@pytest.fixture(scope='module')
def legs():
    cache = os.path.join(ROOT, 'gpurun_cache', 'stage')         # CPU legs computed elsewhere, when the directory travelled
    # 8 workers x 8 threads: the full-size parity suite's pool (tests/test_fullsize_parity_gpu.py) runs at the same time, and
    # fp64 GEMM threads beyond the physical cores only slow both down
    pool = sp.StageLegs(workers=8, threads=8, cache=cache if os.path.isdir(cache) else None)
    pool.submit([(cfg, 'ctor326', kind, SEED, i) for cfg, kind in CASES for i in range(sp.pf.RUNS[cfg][3])])
    yield pool
    pool.close()


@pytest.mark.parametrize('cfg,kind', CASES)
def test_every_stage_within_twice_the_fp32_port(legs, cfg, kind):
    rows = sp.run_case(legs, cfg, 'ctor326', kind, SEED)
    assert len(rows[0]['gpu_mean']) == sp.pf.RUNS[cfg][3] == 4
    bad = [line for r, line in zip(rows, sp.fmt_rows(rows)) if not sp.gate(r)[1]]
    assert not bad, '\n' + '\n'.join(bad)
