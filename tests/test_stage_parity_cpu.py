"""Host logic of the stage-isolated parity harness (tools/stage_parity.py) without a GPU: the CPU oracle is injected as the
"product" backend at a reduced frame size, so every stage's plumbing -- which oracle tap feeds which stage, batch stacking of
the per-sample taps, stream layout of the per-block stages, sign conventions of the stereo path -- is exercised and the backend,
being the fp32 port itself, must land on the port's own error.  The GPU legs proper: tests/test_stage_parity_gpu.py."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import stage_parity as sp  # noqa: E402
from oracle import model as om  # noqa: E402
from oracle_ops import OracleOps, _map  # noqa: E402


class StageOracleOps(OracleOps):
    """OracleOps + the per-scale glue ops the harness calls directly."""

    def flow_upsample2x(self, flow, mult=2.0):
        return F.interpolate(flow, scale_factor=2, mode='bilinear', align_corners=True) * mult

    def flow_warp(self, tokens, flow, h, w):
        return om.warp(_map(tokens, h, w), flow).flatten(2).transpose(1, 2).contiguous()

    def convex_upsample(self, flow, mask, factor, is_depth=False, mask_nhwc=False):
        assert not mask_nhwc
        return om.convex_upsample(flow, mask, factor, is_depth)


@pytest.fixture(scope='module')
def legs():
    """Reduced frame size for the duration of THIS module only: the environment variable reaches the (spawned) worker processes,
    the parent's copy of the size table is patched and restored -- the GPU suites import the same harness at full size."""
    old_env, old_runs = os.environ.get('UM_STAGE_SIZE'), sp.pf.RUNS
    os.environ['UM_STAGE_SIZE'] = '64,96'
    sp.pf.RUNS = {c: (r[0], 64, 96, r[3]) for c, r in old_runs.items()}
    pool = sp.StageLegs(workers=2, threads=2)
    try:
        yield pool
    finally:
        pool.close()
        sp.pf.RUNS = old_runs
        if old_env is None:
            os.environ.pop('UM_STAGE_SIZE', None)
        else:
            os.environ['UM_STAGE_SIZE'] = old_env


@pytest.mark.parametrize('cfg', [3, 4, 2, 5])
def test_harness_with_the_port_as_backend(legs, cfg):
    assert sp.pf.RUNS[cfg][1:3] == (64, 96)
    rows = sp.run_case(legs, cfg, 'ctor326', 'shift', 1000, nsamples=2, backend=StageOracleOps())
    ck, kw, _, _ = sp.pf.case_inputs(cfg, 'shift', 1000)
    want = [(n, k) for n in sp.stage_names(ck, kw) for k, _ in sp.stage_outputs(n, ck, kw)]
    assert [(r['stage'], r['key']) for r in rows] == want
    for r in rows:
        assert len(r['gpu_mean']) == 2
        for g, p, m in zip(r['gpu_mean'], r['port_mean'], r['mag']):
            # same arithmetic up to batching (batch-2 GEMMs / convolutions may round differently from batch 1)
            assert g <= 3.0 * p + 1e-5 * max(m, 1e-3), (r['stage'], r['key'], g, p, m)
        assert all(torch.isfinite(torch.tensor(r['gpu_max'])))
