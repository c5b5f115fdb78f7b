"""world_size-2 tests of the batch-sharded path on CPU (gloo): sharding bounds, the all-gather with unequal
shards and bidirectional layouts, and the wrapped model against the single-process full-batch result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unimatch_amd.dist import ShardedUniMatch, all_gather_predictions, shard_bounds


def test_shard_bounds_cover_the_batch():
    for batch in (1, 2, 3, 8, 13, 32):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # ---- raw gather: unequal shards (batch 3 over 2 ranks), plain and bidirectional layouts
        full = torch.arange(3 * 2 * 4 * 5, dtype=torch.float32).reshape(3, 2, 4, 5)
        lo, hi = shard_bounds(3, rank, world)
        got = all_gather_predictions(full[lo:hi].contiguous(), 3, rank, world)
        ok1 = torch.equal(got, full)
        fwd, bwd = full, -full
        local = torch.cat([fwd[lo:hi], bwd[lo:hi]], 0)
        got = all_gather_predictions(local, 3, rank, world, parts=2)
        ok2 = torch.equal(got, torch.cat([fwd, bwd], 0))

        # ---- the wrapped model (oracle injected as hot-path backend) vs the full batch in one process
        from tests.oracle_ops import OracleOps
        from tests.test_host_logic_cpu import build
        model, sd, i0, i1, kw, ck = build('gmflow_s1', batch=3)
        model.bind_ops(OracleOps())
        want = model(i0, i1, **kw)['flow_preds'][0]
        got = ShardedUniMatch(model)(i0, i1, **kw)['flow_preds'][0]
        # different per-rank batch sizes change the BLAS summation order: compare at the fp32 noise level
        ok3 = got.shape == want.shape and (got - want).abs().mean().item() < 1e-3
        want = model(i0, i1, pred_bidir_flow=True, **kw)['flow_preds'][0]
        got = ShardedUniMatch(model)(i0, i1, pred_bidir_flow=True, **kw)['flow_preds'][0]
        ok4 = got.shape == want.shape == (6, 2, 64, 96) and (got - want).abs().mean().item() < 1e-3
        model, sd, i0, i1, kw, ck = build('gmdepth_s1', batch=2)
        model.bind_ops(OracleOps())
        want = model(i0, i1, **kw)['flow_preds'][0]
        got = ShardedUniMatch(model)(i0, i1, **kw)['flow_preds'][0]
        ok5 = got.shape == want.shape and (got - want).abs().mean().item() < 1e-4
        q.put((rank, ok1, ok2, ok3, ok4, ok5))
    finally:
        dist.destroy_process_group()


def test_world_size_two_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1]
    for r in results:
        assert all(r[1:]), r
