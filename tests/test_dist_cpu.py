"""world_size-2 gloo test of the batch-sharding helpers (the N>1 path of bench.py)."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from global_flow_local_attention_amd import dist as gd
    import torch.distributed as dist
    r, w, _ = gd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    full = torch.arange(7 * 3 * 2 * 2, dtype=torch.float32).view(7, 3, 2, 2)   # ragged: 7 samples on 2 ranks
    mine = gd.shard_batch(full, r, w)
    assert mine.size(0) == (4 if r == 0 else 3)
    got = gd.all_gather_tiles(mine * 1.0)
    assert torch.equal(got, full)
    even = gd.all_gather_tiles(gd.shard_batch(full[:6], r, w))                 # equal shards: single collective
    assert torch.equal(even, full[:6])
    lin = torch.nn.Linear(4, 3)
    torch.manual_seed(0)
    lin.weight.data.fill_(0.5)
    lin.bias.data.zero_()
    x = torch.ones(2, 4) * (r + 1)
    lin(x).sum().backward()
    gd.allreduce_grads(lin.parameters(), average=True)
    assert torch.allclose(lin.weight.grad, torch.full((3, 4), 3.0))            # (2*1 + 2*2)/2
    assert torch.allclose(lin.bias.grad, torch.full((3,), 2.0))
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = True


def test_shard_gather_allreduce_world2():
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1)


def test_shard_range_covers_everything():
    from global_flow_local_attention_amd.dist import shard_range
    for total in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
