"""Batch-sharded multi-GPU inference: one process per GPU, RCCL (``backend='nccl'`` on ROCm) over xGMI.

The hot path shards naturally: every op is per sample (attention windows never cross batch entries and the
[f0;f1] / [f1;f0] stream pairing stays inside a rank as long as a rank holds whole image pairs), so there is
NO collective on the data path.  The only exchange is collecting the predictions: one all-gather of
``[B/N, 2, H, W]`` fp32 per forward (3.1 MB per pair at 512x768) on the compute stream.  The reference has
no inference-time parallelism at all (its only collective is DDP's gradient all-reduce in training,
main_flow.py:188-191); its process-group bring-up role (utils/dist_utils.py:12-30) is taken by
``init_distributed`` below.
"""
import ctypes
import os
import socket
import subprocess
import sys
import warnings
import zlib

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Bring up the process group from the launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world, device).  No-op for a single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    use_gpu = torch.cuda.is_available()
    device = torch.device('cuda', local_rank) if use_gpu else torch.device('cpu')
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = backend or ('nccl' if use_gpu else 'gloo')
        kwargs = {'device_id': device} if backend == 'nccl' else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, device


class RcclGather:
    """The prediction all-gather through the library's own C ABI (``um_allgather_preds`` = ``ncclAllGather`` on the
    caller's stream, include/unimatch_hip.h) -- ``torch.distributed`` is then only the launcher.

    Bootstrap: ``id_file`` (torch-free: rank 0 publishes the 128-byte unique id in a file every rank of the node can
    read), else the id travels through the already initialised ``torch.distributed`` group (any backend)."""

    def __init__(self, rank, world, device, id_file=None, timeout=120):
        from . import _abi
        self._abi, self.lib = _abi, _abi.load()
        self.rank, self.world, self.device = rank, world, torch.device(device)
        self.comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):                       # ncclCommInitRank binds the current device
            if id_file:
                # the record carries a per-job nonce derived from values every rank shares (UM_RCCL_NONCE, else the rendezvous
                # address and the path itself): a record a crashed job left at the same path within the freshness window is not
                # this job's and is ignored by the readers.  Every rank -- rank 0 included -- gives up after `timeout` seconds.
                code = self.lib.um_comm_init_file_nonce(ctypes.byref(self.comm), os.fsencode(id_file), rank, world, timeout,
                                                        job_nonce(id_file))
            else:
                # Rank 0 ALWAYS broadcasts -- the id, or the reason it could not get one -- so that every rank passes through
                # the same collective and then raises the same error (a rank-0-only exception in front of the broadcast left
                # the other ranks blocked in it).
                uid = (ctypes.c_ubyte * _abi.COMM_ID_BYTES)()
                box = [None]
                if rank == 0:
                    try:
                        _abi.check(self.lib.um_comm_unique_id(uid), 'um_comm_unique_id')
                        box = [bytes(uid)]
                    except Exception as exc:                        # noqa: BLE001 -- travels to every rank
                        box = [('error', f'{type(exc).__name__}: {exc}')]
                if world > 1:
                    dist.broadcast_object_list(box, src=0)
                if not isinstance(box[0], bytes):
                    raise RuntimeError(f'RcclGather: rank 0 could not create the RCCL unique id ({box[0][1] if box[0] else "?"})')
                uid = (ctypes.c_ubyte * _abi.COMM_ID_BYTES).from_buffer_copy(box[0])
                code = self.lib.um_comm_init_rank(ctypes.byref(self.comm), uid, rank, world)
        _abi.check(code, 'um_comm_init')

    def ranks(self):
        return self.lib.um_comm_world(self.comm)

    def all_gather(self, send, recv=None, stream=None):
        """send: contiguous fp32 CUDA tensor; returns ``[world, *send.shape]`` (enqueued on ``stream`` / the current one)."""
        assert send.is_cuda and send.dtype == torch.float32 and send.is_contiguous()
        if recv is None:
            recv = torch.empty(self.world, *send.shape, dtype=torch.float32, device=send.device)
        assert recv.is_contiguous() and recv.numel() == self.world * send.numel()
        st = stream if stream is not None else torch.cuda.current_stream(send.device)
        with torch.cuda.device(send.device):
            self._abi.check(self.lib.um_allgather_preds(self.comm, send.data_ptr(), recv.data_ptr(), send.numel(),
                                                        st.cuda_stream), 'um_allgather_preds')
        return recv

    def close(self):
        if self.comm:
            self.lib.um_comm_destroy(self.comm)
            self.comm = ctypes.c_void_p()


class TorchGather:
    """Same interface on the launcher's own process group (``torch.distributed`` all-gather, RCCL underneath on GPUs):
    what ``bench.py`` falls back to -- and says so in its output line -- when the library's communicator cannot be built."""

    def __init__(self, rank, world, device):
        self.rank, self.world, self.device = rank, world, torch.device(device)

    def ranks(self):
        return dist.get_world_size()

    def all_gather(self, send, recv=None, stream=None):
        if recv is None:
            recv = torch.empty(self.world, *send.shape, dtype=send.dtype, device=send.device)
        st = stream if stream is not None else torch.cuda.current_stream(send.device)
        with torch.cuda.stream(st):
            dist.all_gather_into_tensor(recv.view(self.world * send.shape[0], *send.shape[1:]), send)
        return recv

    def close(self):
        pass


def _agree(ok, world, device):
    """MIN over the ranks of a local success flag (the launcher's group); identity for a single rank."""
    if world > 1 and dist.is_initialized():
        flag = torch.tensor([int(ok)], device=device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())
    return bool(ok)


def _preflight(device):
    """Everything of the RCCL bootstrap that can fail on ONE rank before its first collective: the library loads and the
    device can be made current."""
    from . import _abi
    _abi.load()
    with torch.cuda.device(torch.device(device)):
        pass


def make_gather(rank, world, device, id_file=None):
    """The data-path collective of a multi-rank job: ``(gather, description)``.  Tries the library's own communicator
    (``RcclGather``).  Ranks never end up on mixed collectives (those hang): (1) every rank-local step that can fail before the
    bootstrap's broadcast is tried first and the ranks agree on it (MIN all-reduce on the launcher's group), so no rank enters
    the broadcast alone; (2) rank 0 always broadcasts the id or its error; (3) the ranks agree on the outcome and fall back
    TOGETHER to ``TorchGather``, with a warning.  Without a launcher group (``id_file`` bootstrap, no ``torch.distributed``)
    there is nothing to agree through: a failure is raised instead of silently diverging."""
    kind = 'um_allgather_preds (ncclAllGather through the C ABI, own communicator)'
    can_agree = world == 1 or dist.is_initialized()
    gather, why = None, None
    try:
        _preflight(device)
    except Exception as exc:                                        # noqa: BLE001 -- reported in the description
        why = f'{type(exc).__name__}: {exc}'
    if not _agree(why is None, world, device):
        why = why or 'preflight failed on another rank'
    else:
        try:
            gather = RcclGather(rank, world, device, id_file=id_file)
        except Exception as exc:                                    # noqa: BLE001
            why = f'{type(exc).__name__}: {exc}'
            if not can_agree:
                raise
        if not _agree(gather is not None, world, device):
            if gather is not None:
                gather.close()
                gather = None
            why = why or 'um_comm_init failed on another rank'
    if gather is None:
        if not can_agree:
            raise RuntimeError(f'make_gather: no RCCL communicator ({why}) and no torch.distributed group to fall back on')
        kind = f'torch.distributed all_gather_into_tensor (library communicator unavailable: {why})'[:240]
        warnings.warn(f'unimatch_amd.dist: rank {rank} falls back to {kind}')
        gather = TorchGather(rank, world, device)
    return gather, kind


_GATHER = None


def rccl_gather(device=None):
    """The process-wide prediction gather of an initialised multi-rank job on GPUs (created on first use; the library's RCCL
    communicator, or -- agreed by all ranks -- ``torch.distributed`` when that cannot be built)."""
    global _GATHER, GATHER_KIND
    if _GATHER is None:
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        _GATHER, GATHER_KIND = make_gather(rank, world, dev, id_file=os.environ.get('UM_RCCL_ID_FILE'))
    return _GATHER


GATHER_KIND = None          # description of the collective rccl_gather() settled on (make_gather's second result)


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(script, script_args, nproc, need_gpus=True, env=None, capture=False):
    """Run ``script`` as ``nproc`` ranks of one node (what ``python bench.py --gpus N`` does when it was not started
    by a launcher): ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1``.
    Fails loudly -- never silently measures fewer GPUs -- when the node has fewer than ``nproc`` GPUs.
    Returns the launcher's exit code (non-zero as soon as ANY rank exits non-zero: torch.distributed.run tears the others down);
    with ``capture`` it returns ``(exit code, stdout, stderr)`` of the whole job instead of letting the ranks write to the caller's
    streams, so that the caller can refuse to pass on a result line of a job in which a rank failed."""
    if need_gpus:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < nproc:
            raise SystemExit(f'{os.path.basename(script)}: {nproc} GPUs requested but this node exposes {have}')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), script] + list(script_args)
    e = dict(os.environ if env is None else env)
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')                 # dmabuf IPC only on these hosts (RCCL needs it)
    if capture:
        done = subprocess.run(cmd, env=e, capture_output=True, text=True)
        return done.returncode, done.stdout, done.stderr
    return subprocess.call(cmd, env=e)


def job_id_file(port=None):
    """A job-unique path for ``um_comm_init_file`` (rendezvous port + the CALLER's parent pid), for a launcher that bootstraps the
    communicator without ``torch.distributed``: compute it ONCE in the launcher and export it to every rank as UM_RCCL_ID_FILE
    (ranks must not call this themselves unless they share a parent).  ``launch_ranks`` never needs one: its ranks bootstrap through
    the launcher's process group."""
    import tempfile
    port = port if port is not None else os.environ.get('MASTER_PORT', '0')
    return os.path.join(tempfile.gettempdir(), f'um_rccl_id_{port}_{os.getppid()}')


def job_nonce(id_file=None):
    """31-bit tag stored in the id-file record and checked by its readers, so that a relaunch never joins the record a crashed job
    left under the same path.  Built ONLY from values every rank of a job shares however the ranks were started (separate shells,
    per-node agents, a scheduler): ``UM_RCCL_NONCE`` when set (export a fresh value per launch for the strongest guarantee), else a
    hash of MASTER_ADDR:MASTER_PORT and the id-file path.  (Round 4 mixed in ``os.getppid()``: ranks with different parents then
    disagreed and the bootstrap timed out.)"""
    if os.environ.get('UM_RCCL_NONCE'):
        return int(os.environ['UM_RCCL_NONCE']) & 0x7fffffff
    key = f"{os.environ.get('MASTER_ADDR', '')}:{os.environ.get('MASTER_PORT', '0')}:{os.path.abspath(id_file) if id_file else ''}"
    return (zlib.crc32(key.encode()) & 0x7fffffff) or 1


def shard_bounds(batch, rank, world):
    """Contiguous split of ``batch`` samples: rank r owns [lo, hi).  The first ``batch % world`` ranks get one
    extra sample, so any batch size works."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensor, rank, world):
    if tensor is None:
        return None
    lo, hi = shard_bounds(tensor.shape[0], rank, world)
    return tensor[lo:hi].contiguous()


def all_gather_predictions(local, batch, rank, world, parts=1, force=False):
    """Collect per-rank predictions into the full batch on every rank (``force``: run the collective at world size 1 too).

    local: ``[parts * b_r, ...]`` where b_r is this rank's share of ``batch``; ``parts`` = 2 for bidirectional
    outputs, whose layout is [forward(all samples); backward(all samples)] (unimatch.py:139-141) and must be
    rebuilt in that order.  Ranks may own different sample counts: shards are padded to the largest share for
    the collective and trimmed afterwards.
    """
    if world == 1 and not force:
        return local
    counts = [shard_bounds(batch, r, world)[1] - shard_bounds(batch, r, world)[0] for r in range(world)]
    bmax = max(counts)
    tail = local.shape[1:]
    mine = local.reshape(parts, counts[rank], *tail)
    if counts[rank] < bmax:
        pad = torch.zeros(parts, bmax - counts[rank], *tail, dtype=local.dtype, device=local.device)
        mine = torch.cat([mine, pad], 1)
    mine = mine.contiguous()
    if mine.is_cuda and mine.dtype == torch.float32:
        out = rccl_gather(mine.device).all_gather(mine)            # the library's own ncclAllGather, current stream
    else:                                                           # CPU tensors (gloo tests): the launcher's group
        out = torch.empty(world * parts, bmax, *tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, mine)      # concatenation along dim 0, rank major
    out = out.view(world, parts, bmax, *tail)
    pieces = [out[r, :, :counts[r]] for r in range(world)]           # each [parts, b_r, ...]
    full = torch.cat(pieces, 1)                                       # [parts, batch, ...]
    return full.reshape(parts * batch, *tail)


class ShardedUniMatch(torch.nn.Module):
    """Wrap a ``UniMatch`` so that every rank feeds the FULL batch and receives the FULL prediction, while
    computing only its own shard.  With world_size 1 it is the identity wrapper."""

    def __init__(self, model, rank=None, world=None, force_gather=False):
        """``force_gather``: take the sharded path -- shard, forward, ``um_allgather_preds`` -- even with one rank (the
        all-gather degenerates to a copy through RCCL): how a 1-GPU box exercises exactly what N ranks run."""
        super().__init__()
        self.model = model
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.force_gather = force_gather

    def forward(self, img0, img1, **kw):
        batch = img0.shape[0]
        r, n = self.rank, self.world
        if n == 1 and not self.force_gather:
            return self.model(img0, img1, **kw)
        if batch < n:
            raise ValueError(f'batch {batch} is smaller than the number of ranks {n}')
        kw = dict(kw)
        for key in ('intrinsics', 'pose'):
            if kw.get(key) is not None:
                kw[key] = shard_batch(kw[key], r, n)
        local = self.model(shard_batch(img0, r, n), shard_batch(img1, r, n), **kw)['flow_preds'][0]
        parts = 2 if (kw.get('pred_bidir_flow') or kw.get('pred_bidir_depth')) else 1
        return {'flow_preds': [all_gather_predictions(local.contiguous(), batch, r, n, parts, force=self.force_gather)]}
