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


def _id_record(world, nonce, token=0x1122334455667788, magic=b'UMRCCL04', age_ms=0, ttl_ms=2000):
    """A published id record as rank 0 writes it (csrc/rccl_gather.hip IdRecord): magic | world | nonce | stamp ms | deadline ms |
    token | id."""
    import struct
    import time
    now = int(time.time() * 1000) - age_ms
    return magic + struct.pack('<iiqqQ', world, nonce, now, now + ttl_ms, token) + bytes(128)


def test_id_file_readers_skip_records_of_another_job(tmp_path):
    """A record a crashed job left at the path (right magic, right world, fresh stamp) carries that job's nonce: a reader of
    another job must NOT take it -- it times out with a message naming its own nonce instead of joining a dead id."""
    import ctypes
    import time
    from unimatch_amd import _abi
    lib = _abi.load()
    path = tmp_path / 'id'
    path.write_bytes(_id_record(2, 1234))
    comm = ctypes.c_void_p()
    t0 = time.time()
    rc = lib.um_comm_init_file_nonce(ctypes.byref(comm), os.fsencode(str(path)), 1, 2, 1, 4321)
    assert rc == -5 and b'nonce 4321' in lib.um_last_error_string() and time.time() - t0 < 10
    assert path.exists()                                    # a reader never removes the record
    assert not list(tmp_path.glob('id.ack*'))               # ... and does not acknowledge a record that is not its job's
    # a record of an earlier layout (magic 03 / 02) is not taken either, whatever its nonce
    for magic in (b'UMRCCL03', b'UMRCCL02'):
        path.write_bytes(_id_record(2, 4321, magic=magic))
        assert lib.um_comm_init_file_nonce(ctypes.byref(comm), os.fsencode(str(path)), 1, 2, 1, 4321) == -5
    # ... nor one whose shared deadline has passed (a job that died: its readers would wait for a go that never comes)
    path.write_bytes(_id_record(2, 4321, age_ms=5000, ttl_ms=1000))
    assert lib.um_comm_init_file_nonce(ctypes.byref(comm), os.fsencode(str(path)), 1, 2, 1, 4321) == -5
    assert not list(tmp_path.glob('id.ack*'))


def test_id_file_reader_acknowledges_and_gives_up_without_the_go(tmp_path):
    """The three-step rendezvous (publish / ack / go): a reader that finds ITS job's record acknowledges it -- in a file that carries
    the record's per-publish token -- and then waits for rank 0's go until the record's SHARED deadline (not its own clock: ADVICE
    r05); when another rank never shows up rank 0 never gives it, and the reader leaves an abort marker, withdraws its ack and
    returns UM_ERR_COLLECTIVE -- nobody is left inside ncclCommInitRank.  A `.go` of ANOTHER publish (a crashed job under the same
    path and nonce) is not this rendezvous' go."""
    import ctypes
    import time
    from unimatch_amd import _abi
    lib = _abi.load()
    path = tmp_path / 'id'
    token = 0xabcdef0123456789
    path.write_bytes(_id_record(3, 77, token=token, ttl_ms=1500))
    (tmp_path / 'id.go').write_bytes(b'')                                # round-5 name of a stale go
    (tmp_path / f'id.go.{0x1111111111111111:016x}').write_bytes(b'')     # the go of another publish
    comm = ctypes.c_void_p()
    t0 = time.time()
    rc = lib.um_comm_init_file_nonce(ctypes.byref(comm), os.fsencode(str(path)), 2, 3, 30, 77)     # own timeout 30 s: the record's 1.5 s rule
    assert rc == -5 and b"rank 0's go" in lib.um_last_error_string() and 1.0 < time.time() - t0 < 10
    assert not comm.value and not list(tmp_path.glob('id.ack2*'))       # its acknowledgement is withdrawn
    assert (tmp_path / f'id.abort2.{token:016x}').exists()               # ... and it says so


def test_id_file_reader_ignores_a_dead_jobs_go_and_follows_the_new_publish(tmp_path):
    """a job that died inside ncclCommInitRank leaves its record AND its go behind, still fresh.  A reader of the
    relaunch (same path, same nonce) that arrives before the new rank 0 acknowledges the stale record but must not take the stale go
    (it is older than the reader's ack); when rank 0's new record replaces the stale one the reader withdraws its ack and
    acknowledges the new token."""
    import ctypes
    import threading
    import time
    from unimatch_amd import _abi
    lib = _abi.load()
    path = tmp_path / 'id'
    stale, fresh = 0x0101010101010101, 0x0202020202020202
    path.write_bytes(_id_record(2, 55, token=stale, ttl_ms=60000))
    (tmp_path / f'id.go.{stale:016x}').write_bytes(b'')                  # the dead job's go
    time.sleep(0.05)
    out = {}

    def reader():
        comm = ctypes.c_void_p()
        out['rc'] = lib.um_comm_init_file_nonce(ctypes.byref(comm), os.fsencode(str(path)), 1, 2, 30, 55)
        out['err'] = lib.um_last_error_string()
    th = threading.Thread(target=reader)
    th.start()
    time.sleep(0.5)
    assert th.is_alive()                                                 # it did NOT run into ncclCommInitRank on the stale go
    assert (tmp_path / f'id.ack1.{stale:016x}').exists()
    path.write_bytes(_id_record(2, 55, token=fresh, ttl_ms=1000))        # the relaunch's rank 0 publishes (and then never gives a go)
    time.sleep(0.4)
    assert (tmp_path / f'id.ack1.{fresh:016x}').exists() and not (tmp_path / f'id.ack1.{stale:016x}').exists()
    th.join(timeout=10)
    assert not th.is_alive() and out['rc'] == -5 and b"rank 0's go" in out['err']
    assert (tmp_path / f'id.abort1.{fresh:016x}').exists()


def test_id_file_rank0_cleans_stale_files_and_refuses_late_go(tmp_path):
    """Rank 0 removes every auxiliary file an earlier job left under the path before it publishes, publishes a record with a fresh
    token and the shared deadline, and -- alone in a world of 2 -- gives up a margin BEFORE that deadline without ever writing a go."""
    import ctypes
    import struct
    import threading
    import time
    from unimatch_amd import _abi
    lib = _abi.load()
    path = tmp_path / 'id'
    for name in ('id.go', 'id.ack1', 'id.go.00000000deadbeef', 'id.ack1.00000000deadbeef', 'id.abort1.00000000deadbeef'):
        (tmp_path / name).write_bytes(b'')
    seen = {}

    def watch():                                                   # what rank 0 published, read while it waits
        t_end = time.time() + 5
        while time.time() < t_end and 'rec' not in seen:
            try:
                raw = path.read_bytes()
                if len(raw) == 168:
                    seen['rec'] = raw
                    seen['files'] = sorted(p.name for p in tmp_path.iterdir())
            except OSError:
                pass
            time.sleep(0.01)
    warm = (ctypes.c_ubyte * 128)()
    if lib.um_comm_unique_id(warm) != 0:                           # (also loads RCCL once: the load is not part of the timed wait)
        pytest.skip('no RCCL on this box: rank 0 cannot mint an id')
    th = threading.Thread(target=watch)
    th.start()
    comm = ctypes.c_void_p()
    t0 = time.time()
    rc = lib.um_comm_init_file_nonce(ctypes.byref(comm), os.fsencode(str(path)), 0, 2, 2, 99)
    dt = time.time() - t0
    th.join()
    if b'RCCL unavailable' in lib.um_last_error_string():
        pytest.skip('no RCCL on this box: rank 0 cannot mint an id')
    assert rc == -5 and b'acknowledge' in lib.um_last_error_string() and dt < 2.0      # deadline 2 s - margin 0.5 s
    magic, world, nonce, stamp, deadline, token = struct.unpack('<8siiqqQ', seen['rec'][:40])
    assert magic == b'UMRCCL04' and (world, nonce) == (2, 99) and deadline - stamp == 2000 and token not in (0, 0xdeadbeef)
    assert seen['files'] == ['id'], seen['files']                  # the stale files were gone before the record appeared
    assert not list(tmp_path.iterdir())                            # and rank 0 cleaned up after giving up


def test_job_nonce_is_shared_by_ranks_with_different_parents(tmp_path, monkeypatch):
    """The nonce is a function of what the ranks share (UM_RCCL_NONCE, or the rendezvous address and the id-file path) and of nothing
    else: two ranks started by different parent processes compute the same value (getppid made them disagree)."""
    import subprocess
    import sys
    from unimatch_amd import dist as umd
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', '29511')
    monkeypatch.delenv('UM_RCCL_NONCE', raising=False)
    here = umd.job_nonce(str(tmp_path / 'id'))
    code = f"import sys; sys.path.insert(0, {ROOT!r}); from unimatch_amd import dist; print(dist.job_nonce({str(tmp_path / 'id')!r}))"
    # the same question from a grandchild (another parent pid) and through a shell
    other = subprocess.run([sys.executable, '-c', f"import subprocess, sys; print(subprocess.run([sys.executable, '-c', {code!r}], "
                            "capture_output=True, text=True).stdout)"], capture_output=True, text=True, env=dict(os.environ)).stdout
    assert int(other.strip()) == here and here != umd.job_nonce(str(tmp_path / 'other'))
    monkeypatch.setenv('UM_RCCL_NONCE', '12345')
    assert umd.job_nonce('anything') == 12345


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


def test_launcher_reports_a_failed_rank(tmp_path):
    """One rank exiting non-zero makes the whole launch non-zero, and its stderr reaches the caller (``capture``)."""
    from unimatch_amd.dist import launch_ranks
    script = tmp_path / 'ranks.py'
    script.write_text(textwrap.dedent("""
        import os, sys
        if os.environ['RANK'] == '1':
            sys.stderr.write('rank 1: the GPU fell off the bus\\n')
            sys.exit(3)
        print('{"value": 123.0}')
    """))
    rc, out, err = launch_ranks(str(script), [], 2, need_gpus=False, capture=True)
    assert rc != 0 and 'the GPU fell off the bus' in err


@pytest.mark.skipif(torch.cuda.is_available(), reason='drives the ranks into their own no-GPU error')
def test_bench_launcher_never_prints_a_number_when_a_rank_failed(capsys):
    """`python bench.py --gpus 2` as its own launcher: when a rank exits non-zero (here: every rank finds no GPU) the parent prints a
    line with value null, the exit code and the ranks' stderr -- never rank 0's number."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    rc = bench.run_as_launcher(['--gpus', '2', '--steps', '1', '--warmup', '0'], 2, need_gpus=False)
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rc != 0 and line['value'] is None and line['exit_code'] == rc and line['n_gpus'] == 2
    assert 'needs GPU' in line['stderr_tail']


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
