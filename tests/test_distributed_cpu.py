"""world_size-2 tests of the batch-sharded path on CPU (gloo): sharding bounds, the all-gather with unequal
shards and bidirectional layouts, and the wrapped model against the single-process full-batch result."""
import os
import socket
import subprocess
import sys
import textwrap

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


def _gather_worker(rank, world, port, q):
    """make_gather on a box without GPU / RCCL: the library communicator cannot be built on any rank, every rank must learn that
    through the SAME collectives (broadcast of rank 0's outcome, MIN all-reduce of the agreement flag) and fall back together."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from unimatch_amd.dist import TorchGather, make_gather
        gather, kind = make_gather(rank, world, torch.device('cpu'))
        ok = isinstance(gather, TorchGather) and 'torch.distributed' in kind
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)                                   # the group is still in step after the fallback
        q.put((rank, ok, t.item() == 3.0))
    finally:
        dist.destroy_process_group()


def _rank0_failure_worker(rank, world, port, q):
    """Past the preflight (device context patched out), um_comm_unique_id fails on RANK 0 ONLY: the error must reach rank 1 through
    the bootstrap's broadcast (rank 1 must not sit in it alone), both ranks must agree and fall back together, with a warning."""
    import contextlib
    import warnings
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from unimatch_amd import _abi
        from unimatch_amd import dist as umd
        torch.cuda.device = lambda *_a, **_k: contextlib.nullcontext()          # this process only: let the bootstrap reach RCCL
        lib = _abi.load()
        calls = {'uid': 0, 'init': 0}

        class Patched:
            def __getattr__(self, name):
                if name == 'um_comm_unique_id':
                    def fail(_buf):
                        calls['uid'] += 1
                        return -5                                                # UM_ERR_COLLECTIVE
                    return fail
                if name == 'um_comm_init_rank':
                    def init(*_a):
                        calls['init'] += 1
                        return -5
                    return init
                return getattr(lib, name)
        _abi._lib = Patched()
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            gather, kind = umd.make_gather(rank, world, torch.device('cpu'))
        ok = isinstance(gather, umd.TorchGather) and 'unavailable' in kind and any('falls back' in str(w.message) for w in caught)
        ok = ok and calls['uid'] == (1 if rank == 0 else 0) and calls['init'] == 0      # nobody tried to join a communicator
        if rank == 1:
            ok = ok and 'rank 0 could not create' in kind                        # the reason travelled through the broadcast
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)                                                       # the group is still in step
        q.put((rank, ok, t.item() == 3.0))
    finally:
        dist.destroy_process_group()


def test_rank0_only_bootstrap_failure_reaches_every_rank():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank0_failure_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1] and all(r[1] and r[2] for r in results), results


def test_id_file_readers_skip_records_of_another_job(tmp_path):
    """A record a crashed job left at the path (right magic, right world, fresh stamp) carries that job's nonce: a reader of
    another job must NOT take it -- it times out with a message naming its own nonce instead of joining a dead id."""
    import ctypes
    import struct
    import time
    from unimatch_amd import _abi
    lib = _abi.load()
    path = tmp_path / 'id'
    path.write_bytes(b'UMRCCL02' + struct.pack('<iiq', 2, 1234, int(time.time())) + bytes(128))
    comm = ctypes.c_void_p()
    t0 = time.time()
    rc = lib.um_comm_init_file_nonce(ctypes.byref(comm), os.fsencode(str(path)), 1, 2, 1, 4321)
    assert rc == -5 and b'nonce 4321' in lib.um_last_error_string() and time.time() - t0 < 10
    assert path.exists()                                    # a reader never removes the record


@pytest.mark.skipif(torch.cuda.is_available(), reason='exercises the no-GPU fallback agreement')
def test_gather_bootstrap_falls_back_on_every_rank_together():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1] and all(r[1] and r[2] for r in results), results


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


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_starts_every_rank(tmp_path):
    """``launch_ranks`` -- what ``python bench.py --gpus N`` uses when no launcher started it -- runs N ranks under
    torch.distributed.run with the rendezvous on 127.0.0.1 (driven here on CPU with a gloo all-reduce)."""
    from unimatch_amd.dist import launch_ranks
    script = tmp_path / 'ranks.py'
    script.write_text(textwrap.dedent(f"""
        import os, sys, torch, torch.distributed as dist
        dist.init_process_group('gloo')
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        open(os.path.join(r'{tmp_path}', 'rank%d.txt' % dist.get_rank()), 'w').write('%d %d %s' % (dist.get_world_size(), int(t.item()), sys.argv[1]))
        dist.destroy_process_group()
    """))
    assert launch_ranks(str(script), ['hello'], 2, need_gpus=False) == 0
    for r in range(2):
        assert (tmp_path / f'rank{r}.txt').read_text() == '2 3 hello'


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2, reason='needs a node with fewer than 2 GPUs')
def test_bench_refuses_to_measure_fewer_gpus_than_asked():
    """`python bench.py --gpus 2` on a node with fewer than two GPUs must fail loudly, not print an n_gpus=1 line; and a
    launcher whose WORLD_SIZE differs from --gpus is an error as well."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert out.returncode != 0 and '2 GPUs requested' in (out.stderr + out.stdout)
    assert '"n_gpus"' not in out.stdout
    env.update(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert out.returncode != 0 and 'WORLD_SIZE=2' in (out.stderr + out.stdout)
