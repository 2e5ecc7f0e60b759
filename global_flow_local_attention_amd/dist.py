"""Batch (data) parallelism for the hot path: one process per GPU, torch.distributed over RCCL.

Every op on the path is per-sample (no index ever mixes batch entries), which is also the only
way the reference parallelises (nn.DataParallel scatter along dim 0, model/face_model.py:81-93).
So ranks own disjoint slices of the batch and run identical kernels; there is no collective inside
the data path.  Two exchanges exist around it:
  * inference: all-gather of the generated tiles when one rank must own the whole batch
    (DataParallel's gather step) -> `all_gather_tiles`;
  * training: sum of the ExtractorAttn parameter gradients across ranks (DataParallel's
    reduce step) -> `allreduce_grads`, one flat bucket so a ring step is link-bound on xGMI.
Backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun's contract).
    Returns (rank, world_size, local_rank).  A single process without those variables is rank 0 of 1
    and does not create a process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total, rank, world):
    """[start, end) of the samples rank owns; remainders go to the lowest ranks."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_batch(tensor, rank, world):
    s, e = shard_range(tensor.size(0), rank, world)
    return tensor[s:e]


def all_gather_tiles(local_tiles, total=None):
    """Concatenate every rank's (b_r, ...) tiles along dim 0 on every rank.  Equal shards use one
    all_gather_into_tensor (a single large transfer per peer); ragged shards are padded."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_tiles
    world = dist.get_world_size()
    local_tiles = local_tiles.contiguous()
    sizes = torch.tensor([local_tiles.size(0)], device=local_tiles.device, dtype=torch.int64)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    counts = [int(s.item()) for s in all_sizes]
    mx = max(counts)
    if min(counts) == mx:
        out = local_tiles.new_empty((world * mx,) + tuple(local_tiles.shape[1:]))
        dist.all_gather_into_tensor(out, local_tiles)
        return out
    pad = local_tiles.new_zeros((mx,) + tuple(local_tiles.shape[1:]))
    pad[:local_tiles.size(0)] = local_tiles
    out = local_tiles.new_empty((world * mx,) + tuple(local_tiles.shape[1:]))
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx:r * mx + counts[r]] for r in range(world)], 0)


def allreduce_grads(params, average=True):
    """Sum (or average) .grad of `params` across ranks through ONE flat bucket."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
