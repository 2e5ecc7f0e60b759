"""trainer.py: reference-checkpoint loading rules and the one-process-per-rank training step (gloo, world size 2)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.ReLU(), torch.nn.Linear(3, 2))


def test_load_reference_checkpoint_rules():
    from global_flow_local_attention_amd.trainer import load_reference_checkpoint
    src = _net()
    with torch.no_grad():
        for p in src.parameters():
            p.add_(1.0)
    # exact
    dst = _net()
    assert load_reference_checkpoint(dst, src.state_dict()) == []
    assert all(torch.equal(a, b) for a, b in zip(dst.state_dict().values(), src.state_dict().values()))
    # DataParallel-style "module." prefixes (base_model.py:174-175)
    dst = _net()
    assert load_reference_checkpoint(dst, {"module." + k: v for k, v in src.state_dict().items()}) == []
    assert torch.equal(dst[0].weight, src[0].weight)
    # excessive layers in the checkpoint: only the used ones are loaded (:171-179)
    dst = _net()
    extra = dict(src.state_dict(), **{"9.weight": torch.zeros(1)})
    assert load_reference_checkpoint(dst, extra) == []
    # fewer / mismatching layers: shape-matching entries load, the rest is reported (:181-192)
    dst = _net()
    partial = {k: v for k, v in src.state_dict().items() if not k.startswith("2.")}
    partial["2.weight"] = torch.zeros(5, 5)
    assert load_reference_checkpoint(dst, partial) == ["2"]
    assert torch.equal(dst[0].weight, src[0].weight) and not torch.equal(dst[2].weight, src[2].weight)


@pytest.mark.skipif(not os.path.isdir("/root/reference/model/networks"), reason="reference checkout not present")
def test_reference_pose_generator_checkpoint_roundtrip(tmp_path):
    """A `*_net_G.pth` written from the reference's own PoseGenerator (also from a DataParallel wrapper) loads into the
    install()-patched generator: same keys, same shapes (generator.py setattr naming, base_function.py:799-803)."""
    import subprocess
    code = r"""
import sys, types, torch
sys.path.insert(0, %r)
import global_flow_local_attention_amd as g
from global_flow_local_attention_amd.trainer import load_reference_checkpoint
sys.modules.setdefault('torchvision', types.ModuleType('torchvision'))
g.install('/root/reference')
import model.networks.generator as gen
kw = dict(image_nc=3, structure_nc=18, ngf=16, img_f=64, layers=3, num_blocks=1, use_spect=False,
          attn_layer=[2, 3], norm='instance', activation='LeakyReLU', extractor_kz={'2': 5, '3': 3})
torch.manual_seed(1); a = gen.PoseGenerator(**kw)
torch.manual_seed(2); b = gen.PoseGenerator(**kw)
torch.save(torch.nn.DataParallel(a).state_dict(), %r)          # keys carry the 'module.' prefix
assert load_reference_checkpoint(b, %r) == []
assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))
assert any(k.endswith('fully_connect_layer.0.weight') for k in b.state_dict())
print('ok')
""" % (ROOT, str(tmp_path / "latest_net_G.pth"), str(tmp_path / "latest_net_G.pth"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


class _TinyGenerator(torch.nn.Module):
    """Stands in for PoseGenerator on the CPU: same (generated, flow_fields, masks) return convention."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(7)
        self.body = torch.nn.Conv2d(3, 3, 3, padding=1)
        self.flow = torch.nn.Conv2d(3, 2, 3, padding=1)

    def forward(self, img):
        return self.body(img), [self.flow(img)], []


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from global_flow_local_attention_amd import dist as gd
    from global_flow_local_attention_amd.losses import MultiAffineRegularizationLoss
    from global_flow_local_attention_amd.trainer import TrainerShell
    import torch.distributed as dist
    gd.init_from_env(backend="gloo")
    net = _TinyGenerator()
    shell = TrainerShell(net, lr=1e-2, regularization=MultiAffineRegularizationLoss({"3": 3}), lambda_regularization=0.01)
    g = torch.Generator().manual_seed(3)
    img, tgt = torch.randn(4, 3, 8, 8, generator=g), torch.randn(4, 3, 8, 8, generator=g)
    mine = shell.shard(img, tgt)
    assert mine[0].size(0) == 2
    losses = shell.optimize_parameters((mine[0],), mine[1])
    assert set(losses) == {"app_gen", "regularization"}
    # every rank holds the same averaged gradient, hence the same updated weights
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.allclose(gathered[0], gathered[1], atol=1e-7)
    ret[rank] = flat.clone()
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_shell_world2_matches_single_process():
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29800 + os.getpid() % 90
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    # single process on the whole batch: mean-reduced losses => the same step as the average of the two shards' gradients
    sys.path.insert(0, ROOT)
    from global_flow_local_attention_amd.losses import MultiAffineRegularizationLoss
    from global_flow_local_attention_amd.trainer import TrainerShell
    net = _TinyGenerator()
    shell = TrainerShell(net, lr=1e-2, regularization=MultiAffineRegularizationLoss({"3": 3}), lambda_regularization=0.01)
    g = torch.Generator().manual_seed(3)
    img, tgt = torch.randn(4, 3, 8, 8, generator=g), torch.randn(4, 3, 8, 8, generator=g)
    shell.optimize_parameters((img,), tgt)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(flat, ret[0], atol=1e-5)


def test_warp_generator_step_on_host_with_oracle_blocks():
    """The generator-shaped network + trainer shell, every hot-path op the oracle's: shapes, conventions
    (generator.py:13-36 return triple, coarse-to-fine flow order), all parameters reached, one Adam step taken."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import trainer_util as tu
    shell, net = tu.build_shell("cpu", ngf=8)
    batch = tu.make_batch(2, 32, 32)
    gen, flows, masks = net(batch[0], batch[2], batch[3])
    assert gen.shape == (2, 3, 32, 32)
    assert [tuple(f.shape) for f in flows] == [(2, 2, 4, 4), (2, 2, 8, 8)]
    assert [tuple(m.shape) for m in masks] == [(2, 1, 4, 4), (2, 1, 8, 8)]
    keys = set(net.state_dict())
    assert {"attn3.fully_connect_layer.0.weight", "attn2.fully_connect_layer.2.bias"} <= keys
    losses, grads, before, after = tu.run_step(shell, net, batch, "cpu")
    assert set(losses) == {"app_gen", "correctness_gen", "regularization"}
    assert all(v == v and abs(v) < 1e6 for v in losses.values())
    assert set(grads) == set(before), set(before) - set(grads)           # every parameter got a gradient
    moved = [n for n in before if not torch.equal(before[n], after[n])]
    assert len(moved) == len(before)
