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
# round 5: the one-scale configs 1 / 2 / 5 (flow at two sizes, depth) on both weight sets, 2 samples of the config's batch each (the full
# batches: tools/stage_parity.py --configs 1,2,5, table in profiles/r05_stage_parity_one_scale.txt).  Stages added for them: the upsampler's
# mask head, the convex combination on the oracle's logits, the depth plane sweep.
ONE_SCALE_SAMPLES = {1: 1, 2: 2, 5: 2}
ONE_SCALE_CASES = [(cfg, which) for cfg in (1, 2, 5) for which in ('ctor326', 'conditioned')]


@pytest.fixture(scope='module')
def legs():
    cache = os.path.join(ROOT, 'gpurun_cache', 'stage')         # CPU legs computed elsewhere, when the directory travelled
    # 8 workers x 8 threads: the full-size parity suite's pool (tests/test_fullsize_parity_gpu.py) runs at the same time, and
    # fp64 GEMM threads beyond the physical cores only slow both down
    pool = sp.StageLegs(workers=8, threads=8, cache=cache if os.path.isdir(cache) else None)
    pool.submit([(cfg, 'ctor326', kind, SEED, i) for cfg, kind in CASES for i in range(sp.pf.RUNS[cfg][3])])
    pool.submit([(cfg, which, 'shift', SEED, i) for cfg, which in ONE_SCALE_CASES for i in range(ONE_SCALE_SAMPLES[cfg])])
    yield pool
    pool.close()


@pytest.mark.parametrize('cfg,kind', CASES)
def test_every_stage_within_twice_the_fp32_port(legs, cfg, kind):
    rows = sp.run_case(legs, cfg, 'ctor326', kind, SEED)
    assert len(rows[0]['gpu_mean']) == sp.pf.RUNS[cfg][3] == 4
    bad = [line for r, line in zip(rows, sp.fmt_rows(rows)) if not sp.gate(r)[1]]
    assert not bad, '\n' + '\n'.join(bad)


@pytest.mark.parametrize('cfg,which', ONE_SCALE_CASES)
def test_one_scale_configs_stage_by_stage(legs, cfg, which):
    """Configs 1, 2, 5: every stage within 2 x the fp32 port on both weight sets.  (Round 5 held the global correlation softmax /
    global propagation rows of the CONDITIONED weights to 4 x: a lane's 3072 softmax terms were one fp32 chain.  Round 6's two-level
    accumulation in gsv4_kernel -- level-2 sums parked in LDS, flushed every 8 key tiles -- removed the exception.)"""
    rows = sp.run_case(legs, cfg, which, 'shift', SEED, nsamples=ONE_SCALE_SAMPLES[cfg])
    names = {r['stage'] for r in rows}
    assert {'encoder', 'xfmr_s0', 'match_s0', 'prop_s0', 'mask_head', 'convex1', 'upsample'} <= names
    bad = []
    for r, line in zip(rows, sp.fmt_rows(rows)):
        worst, ok = sp.gate(r)
        if not ok:
            bad.append(line)
    assert not bad, '\n' + '\n'.join(bad)
