"""Batch-sharded multi-GPU inference: one process per GPU, RCCL (``backend='nccl'`` on ROCm) over xGMI.

The hot path shards naturally: every op is per sample (attention windows never cross batch entries and the
[f0;f1] / [f1;f0] stream pairing stays inside a rank as long as a rank holds whole image pairs), so there is
NO collective on the data path.  The only exchange is collecting the predictions: one all-gather of
``[B/N, 2, H, W]`` fp32 per forward (3.1 MB per pair at 512x768) on the compute stream.  The reference has
no inference-time parallelism at all (its only collective is DDP's gradient all-reduce in training,
main_flow.py:188-191); its process-group bring-up role (utils/dist_utils.py:12-30) is taken by
``init_distributed`` below.
"""
import os

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


def all_gather_predictions(local, batch, rank, world, parts=1):
    """Collect per-rank predictions into the full batch on every rank.

    local: ``[parts * b_r, ...]`` where b_r is this rank's share of ``batch``; ``parts`` = 2 for bidirectional
    outputs, whose layout is [forward(all samples); backward(all samples)] (unimatch.py:139-141) and must be
    rebuilt in that order.  Ranks may own different sample counts: shards are padded to the largest share for
    the collective and trimmed afterwards.
    """
    if world == 1:
        return local
    counts = [shard_bounds(batch, r, world)[1] - shard_bounds(batch, r, world)[0] for r in range(world)]
    bmax = max(counts)
    tail = local.shape[1:]
    mine = local.reshape(parts, counts[rank], *tail)
    if counts[rank] < bmax:
        pad = torch.zeros(parts, bmax - counts[rank], *tail, dtype=local.dtype, device=local.device)
        mine = torch.cat([mine, pad], 1)
    mine = mine.contiguous()
    out = torch.empty(world * parts, bmax, *tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, mine)          # concatenation along dim 0, rank major
    out = out.view(world, parts, bmax, *tail)
    pieces = [out[r, :, :counts[r]] for r in range(world)]           # each [parts, b_r, ...]
    full = torch.cat(pieces, 1)                                       # [parts, batch, ...]
    return full.reshape(parts * batch, *tail)


class ShardedUniMatch(torch.nn.Module):
    """Wrap a ``UniMatch`` so that every rank feeds the FULL batch and receives the FULL prediction, while
    computing only its own shard.  With world_size 1 it is the identity wrapper."""

    def __init__(self, model, rank=None, world=None):
        super().__init__()
        self.model = model
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)

    def forward(self, img0, img1, **kw):
        batch = img0.shape[0]
        r, n = self.rank, self.world
        if n == 1:
            return self.model(img0, img1, **kw)
        if batch < n:
            raise ValueError(f'batch {batch} is smaller than the number of ranks {n}')
        kw = dict(kw)
        for key in ('intrinsics', 'pose'):
            if kw.get(key) is not None:
                kw[key] = shard_batch(kw[key], r, n)
        local = self.model(shard_batch(img0, r, n), shard_batch(img1, r, n), **kw)['flow_preds'][0]
        parts = 2 if (kw.get('pred_bidir_flow') or kw.get('pred_bidir_depth')) else 1
        return {'flow_preds': [all_gather_predictions(local.contiguous(), batch, r, n, parts)]}
