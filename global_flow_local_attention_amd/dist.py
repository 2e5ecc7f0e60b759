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


def init_from_env(backend=None, device=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun's contract).
    Returns (rank, world_size, local_rank).  A single process without those variables is rank 0 of 1
    and does not create a process group.  `device`: the GPU index this rank drives (default LOCAL_RANK); with the
    nccl (= RCCL) backend it is bound before the group exists and handed to init_process_group as `device_id`, so the
    communicator is created eagerly on the right GPU instead of on whatever device the first collective sees."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            dev = local if device is None else int(device)
            torch.cuda.set_device(dev)
            kwargs["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
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
    """Sum (or average) .grad of `params` across ranks through ONE flat bucket, after backward.  A parameter whose
    .grad is None on this rank contributes zeros (and receives the reduced value), so ranks whose graphs touched
    different parameters stay in step."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for p in params:
        n = p.numel()
        if p.grad is None:
            p.grad = flat[off:off + n].view_as(p).clone()
        else:
            p.grad.copy_(flat[off:off + n].view_as(p))
        off += n


class GradBucketReducer(object):
    """DDP-style gradient reduction overlapped with backward: parameters are grouped into flat buckets
    (reverse registration order ~ the order backward produces them); a post-accumulate hook copies each gradient
    into its bucket and, when a bucket is complete, launches its all-reduce asynchronously -- RCCL runs it on its
    own stream over xGMI while the rest of backward still computes.  `finish()` (after backward) waits, averages
    and points every .grad at its slice of the reduced bucket.

    bucket_mb: ring all-reduce over xGMI is per-link bound (7 links x ~153 GB/s per GPU), so buckets are large
    (default 32 MB) -- fewer, longer transfers; the ExtractorAttn parameters of the hot path (5.6 MB) are one bucket.
    Every rank must build the reducer over the same parameters in the same order."""

    def __init__(self, params, bucket_mb=32.0, average=True):
        self.average = average
        self.params = [p for p in params if p.requires_grad]
        self.enabled = dist.is_initialized() and dist.get_world_size() > 1
        self.buckets = []  # {"flat", "params": [(p, offset)], "pending", "work"}
        self._where = {}
        cap = int(bucket_mb * (1 << 20))
        cur = None
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur is None or cur["bytes"] + nbytes > cap or cur["dtype"] != p.dtype or cur["device"] != p.device:
                cur = {"bytes": 0, "dtype": p.dtype, "device": p.device, "params": [], "numel": 0}
                self.buckets.append(cur)
            cur["params"].append((p, cur["numel"]))
            cur["numel"] += p.numel()
            cur["bytes"] += nbytes
        for i, bk in enumerate(self.buckets):
            bk["flat"] = torch.zeros(bk["numel"], dtype=bk["dtype"], device=bk["device"])
            bk["pending"], bk["work"], bk["seen"] = len(bk["params"]), None, set()
            for p, off in bk["params"]:
                self._where[p] = (i, off)
        self._handles = []
        self._next = 0  # first bucket whose all-reduce has not been launched
        if self.enabled:
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _launch(self, bk):
        bk["work"] = dist.all_reduce(bk["flat"], op=dist.ReduceOp.SUM, async_op=True)

    def _launch_ready(self):
        # strictly in bucket order (as DDP does): RCCL matches collectives by ISSUE order, so a rank whose graph
        # completed bucket 1 before bucket 0 must still issue 0 first -- bucket i waits for 0..i-1
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _on_grad(self, p):
        i, off = self._where[p]
        bk = self.buckets[i]
        if p in bk["seen"]:
            # a second backward before finish() (gradient accumulation, retain_graph): p.grad holds the accumulated
            # value, which is what has to be reduced -- fine as long as the bucket has not left yet
            if bk["work"] is not None:
                raise RuntimeError("GradBucketReducer: a parameter received a second gradient after its bucket's "
                                   "all-reduce was launched; call finish() after every backward (or build the reducer "
                                   "after the accumulation steps)")
            bk["flat"][off:off + p.numel()].copy_(p.grad.reshape(-1))
            return
        bk["flat"][off:off + p.numel()].copy_(p.grad.reshape(-1))
        bk["seen"].add(p)
        bk["pending"] -= 1
        self._launch_ready()

    def finish(self):
        """Call after backward: launches what is still pending IN BUCKET ORDER (parameters that got no gradient count
        as zero), waits for every bucket, and leaves the reduced gradients in .grad (views into the buckets)."""
        if not self.enabled:
            return
        world = dist.get_world_size()
        for bk in self.buckets[self._next:]:
            for p, off in bk["params"]:
                if p not in bk["seen"]:  # no gradient in this backward: contributes zeros
                    bk["flat"][off:off + p.numel()].zero_()
            self._launch(bk)
        for bk in self.buckets:
            bk["work"].wait()
            if self.average:
                bk["flat"] /= world
            for p, off in bk["params"]:
                p.grad = bk["flat"][off:off + p.numel()].view_as(p)
            bk["pending"], bk["work"], bk["seen"] = len(bk["params"]), None, set()
        self._next = 0

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
