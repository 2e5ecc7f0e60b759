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


# ------------------------------------------------------------------ overlapped bucketed reducer + bench control flow
def _reducer_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from global_flow_local_attention_amd import dist as gd
    import torch.distributed as dist
    gd.init_from_env(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    unused = torch.nn.Parameter(torch.ones(7))                     # never receives a gradient: counts as zero
    params = list(net.parameters()) + [unused]
    red = gd.GradBucketReducer(params, bucket_mb=100e-6)           # ~100-byte buckets: several buckets, hooks fire mid-backward
    assert len(red.buckets) > 2
    for step in range(2):                                          # second step: bucket state resets
        for p in params:
            p.grad = None
        x = torch.full((2, 6), float(rank + 1 + step))
        net(x).sum().backward()
        red.finish()
        # reference: plain single-process average of the two ranks' gradients
        want = []
        for rr in range(world):
            ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
            ref.load_state_dict(net.state_dict())
            ref(torch.full((2, 6), float(rr + 1 + step))).sum().backward()
            want.append([p.grad.clone() for p in ref.parameters()])
        for i, p in enumerate(net.parameters()):
            assert torch.allclose(p.grad, (want[0][i] + want[1][i]) / 2, atol=1e-6), (step, i)
        assert torch.equal(unused.grad, torch.zeros(7))
    red.remove()
    # ranks whose graphs use DIFFERENT parameters (data-dependent branch): buckets complete in different orders on the
    # two ranks; launches must still be issued in bucket order or the collectives pair up wrongly / hang
    torch.manual_seed(1)
    br = torch.nn.ModuleList([torch.nn.Linear(6, 3), torch.nn.Linear(6, 3), torch.nn.Linear(3, 2)])
    red2 = gd.GradBucketReducer(list(br.parameters()), bucket_mb=60e-6)
    assert len(red2.buckets) >= 3
    x = torch.full((2, 6), float(rank + 1))
    br[2](br[rank](x)).sum().backward()                             # rank r only touches branch r
    red2.finish()
    for rr in range(world):
        ref = torch.nn.ModuleList([torch.nn.Linear(6, 3), torch.nn.Linear(6, 3), torch.nn.Linear(3, 2)])
        ref.load_state_dict(br.state_dict())
        ref[2](ref[rr](torch.full((2, 6), float(rr + 1)))).sum().backward()
        assert torch.allclose(br[rr].weight.grad, ref[rr].weight.grad / 2, atol=1e-6)
        assert torch.allclose(br[rr].bias.grad, ref[rr].bias.grad / 2, atol=1e-6)
    # a second backward after a bucket has left must not go unnoticed
    red2.remove()
    one = torch.nn.Linear(6, 3)
    red3 = gd.GradBucketReducer(list(one.parameters()))               # one bucket, complete after the first backward
    one(x).sum().backward()
    try:
        one(x).sum().backward()
        raised = False
    except RuntimeError as e:
        raised = "second gradient" in str(e)
    assert raised
    red3.finish()
    red3.remove()
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = True


def test_grad_bucket_reducer_world2():
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1)


class _StubHotPath(object):
    """Stands in for bench.HotPath on CPU: same step()/params() surface, a tiny torch model, the same reducer."""

    def __init__(self, rank):
        from global_flow_local_attention_amd import dist as gd
        torch.manual_seed(1234)
        self.net = torch.nn.Linear(8, 4)
        self.x = torch.full((3, 8), float(rank + 1))
        self.reducer = None
        self.gd = gd

    def params(self):
        return list(self.net.parameters())

    def step(self, resample, allreduce=True):
        for p in self.params():
            p.grad = None
        if allreduce and self.reducer is None:
            self.reducer = self.gd.GradBucketReducer(self.params())
        out = resample(self.net(self.x))
        out.sum().backward()
        if allreduce:
            self.reducer.finish()
        return [out]


def _bench_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import bench
    from global_flow_local_attention_amd import dist as gd
    import torch.distributed as dist
    r, w, _ = gd.init_from_env(backend="gloo")
    args = bench.parse_args(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "4", "--no-cpu-baseline"])
    line = bench.run(args, lambda mode: _StubHotPath(r), lambda: (lambda t: t * 2.0), r, w, torch.device("cpu"),
                     on_gpu=False)
    assert line["n_gpus"] == world and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["global_batch"] == 4 * world and line["value"] > 0 and line["scaling"] == "weak"
    # the N > 1 leg the first multi-GPU run will record: process-group fingerprint + the all-gather of generated tiles
    d = line["dist"]
    assert d["backend"] == "gloo" and d["world_size_seen_by_group"] == world
    assert d["allreduce_of_rank_ids"] == d["allreduce_expected"] == world * (world + 1) / 2.0
    ag = d["all_gather_tiles"]
    assert ag["correct"] and ag["tile_shape_per_rank"][0] == 4 and ag["us"] > 0, ag
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = line["value"]


def test_bench_control_flow_world2_gloo():
    """bench.py's N>1 path (barriers, MAX-over-ranks timing, hook-launched all-reduce, rank-0 JSON) on two gloo ranks
    with a stand-in workload: the part of the multi-GPU run that does not need a GPU."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29700 + os.getpid() % 190
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1)
    assert abs(ret[0] - ret[1]) < 1e-6 * ret[0]        # both ranks computed the same MAX-over-ranks time


def test_bench_self_spawn_world2_gloo():
    """`python bench.py --gpus 2` with NO torchrun environment (the shape of the N=1 command with a different N) must not
    exit with a usage error: it spawns its own ranks under torch.distributed.run and rank 0 prints ONE JSON line with
    n_gpus = 2.  GFLA_BENCH_CPU_STUB swaps the GPU workload for a stand-in so the launcher runs where there is no GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GFLA_BENCH_CPU_STUB="1", OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--batch", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 8 and line["value"] > 0
    assert line["dist"]["world_size_seen_by_group"] == 2 and line["dist"]["allreduce_of_rank_ids"] == 3.0
    assert "cpu-stub" in line["data"]


def test_bench_torchrun_form_still_works_gloo():
    """The contract's own launch form (python -m torch.distributed.run ... bench.py --gpus N) must not spawn again."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GFLA_BENCH_CPU_STUB="1", OMP_NUM_THREADS="1")
    port = 29300 + os.getpid() % 190
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "spawning" not in res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
