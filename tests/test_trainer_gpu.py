"""SURVEY 8f row 4 on hardware: one `TrainerShell.optimize_parameters` step (pose_model.py:186-196) of a
generator-SHAPED network -- conv encoders, flow head, ExtractorAttn L3 + L2 with the mask blend, decoder -- with the
sampling-correctness loss (frozen random "VGG") and the affine regulariser, on cuda:0 through this library's kernels,
against the identical step on the host where every hot-path op is the oracle's (oracle/cpu_modules.py).

Bars: every loss term <= 1e-4 relative; every parameter's gradient <= 1e-4 of that tensor's largest entry (fp32 convs /
InstanceNorm on both sides differ by accumulation order); the Adam step really is Adam on the GPU's own gradients, and
agrees with the host's update wherever the gradient is not within noise of zero (Adam's first step is
lr*g/(|g|+eps): a sign-like function, ill-conditioned at g ~ 0 by construction)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEV = "cuda:0"
LR = 1e-3


def _compare_step(ngf, H, W, B):
    import trainer_util as tu
    from global_flow_local_attention_amd import _lib
    batch = tu.make_batch(B, H, W)
    cpu_shell, cpu_net = tu.build_shell("cpu", ngf=ngf, lr=LR)
    state = {k: v.clone() for k, v in cpu_net.state_dict().items()}
    gpu_shell, gpu_net = tu.build_shell(DEV, ngf=ngf, lr=LR, state=state)
    from global_flow_local_attention_amd import fc_mfma
    fm, bm = _lib.fc_path(fc_mfma.DEFAULT_MODE), _lib.fc_path(fc_mfma.DEFAULT_MODE, backward=True)
    fwd0, bwd0 = _lib.path_count(fm), _lib.path_count(bm)
    want_losses, want_grads, before, want_after = tu.run_step(cpu_shell, cpu_net, batch, "cpu")
    losses, grads, _, after = tu.run_step(gpu_shell, gpu_net, batch, DEV)
    # both attention layers went through the default float32 MFMA path (Winograd-domain kernels), forward and backward
    assert _lib.path_count(fm) == fwd0 + 2 and _lib.path_count(bm) == bwd0 + 2
    assert set(losses) == set(want_losses) == {"app_gen", "correctness_gen", "regularization"}
    for k in losses:
        assert abs(losses[k] - want_losses[k]) <= 1e-4 * max(abs(want_losses[k]), 1e-3), (k, losses[k], want_losses[k])
    assert set(grads) == set(want_grads) == set(before)
    worst = ("", 0.0)
    gmax = max(w.abs().max().item() for w in want_grads.values())
    for n in sorted(grads):
        g, w = grads[n].cpu().double(), want_grads[n].double()
        scale = w.abs().max().item()
        # a convolution bias in front of an InstanceNorm has an exactly-zero true gradient (the norm removes the mean): both
        # sides then hold the rounding noise of a cancelling sum (~1e-8 .. 1e-7) -- required to BE noise on both sides
        if scale <= 1e-5 * gmax:
            assert g.abs().max().item() <= 1e-5 * gmax, "grad %s: %.3e where the host has %.3e" % (n, g.abs().max().item(), scale)
            continue
        err = (g - w).abs().max().item() / scale
        assert err <= 1e-4, "grad %s: %.3e of its max %.3e" % (n, err, scale)
        worst = max(worst, (n, err), key=lambda t: t[1])
        # the step taken on the GPU is Adam (betas (0, 0.999), first step) on the GPU's own gradient
        g32 = grads[n].float()
        step = LR * g32 / (g32.abs() + 1e-8)
        assert torch.allclose(after[n], before[n].to(DEV) - step, atol=2e-7 + 1e-3 * LR), n
        # and it is the host's update wherever the gradient is clear of zero
        clear = w.abs() > 1e-2 * scale
        if clear.any():
            d_gpu = (after[n].cpu().double() - before[n].double())[clear]
            d_cpu = (want_after[n].double() - before[n].double())[clear]
            assert (d_gpu - d_cpu).abs().max().item() <= 1e-2 * LR, n
    return worst


def test_trainer_step_on_gpu_matches_host_oracle_step():
    worst = _compare_step(ngf=16, H=64, W=48, B=2)   # L3 (64, 8x6) k3, L2 (32, 16x12) k5
    print("worst gradient: %s %.2e" % worst)


def test_trainer_step_wider_network():
    _compare_step(ngf=32, H=96, W=64, B=1)           # L3 (128, 12x8), L2 (64, 24x16)


def _rank_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import trainer_util as tu
    from global_flow_local_attention_amd import dist as gd
    import torch.distributed as dist
    torch.cuda.set_device(0)                       # both ranks share the one GPU of the box; gloo carries the gradients
    gd.init_from_env(backend="gloo")
    shell, net = tu.build_shell(DEV, ngf=16, lr=LR, bucket_mb=0.25)   # several buckets: hooks fire during backward
    assert len(shell.reducer.buckets) >= 3
    batch = tu.make_batch(4, 64, 48)
    mine = shell.shard(*batch)
    assert mine[0].size(0) == 2
    src, tgt, src_B, tgt_B = (t.to(DEV) for t in mine)
    losses = shell.optimize_parameters((src, src_B, tgt_B), tgt, source=src)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu()
    grad = torch.cat([p.grad.detach().reshape(-1) for p in net.parameters()]).cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1])    # identical weights on both ranks after the step
    ret[rank] = (losses, grad)
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_step_two_ranks_sharing_the_gpu():
    """Two processes (gloo) on the one GPU: the reducer's hooks fire on the full generator's parameters in several
    buckets during a real backward through the HIP kernels; both ranks end with identical weights, and the averaged
    gradient equals the single-process gradient of the whole batch (mean-reduced losses)."""
    import trainer_util as tu
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29300 + os.getpid() % 200
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    shell, net = tu.build_shell(DEV, ngf=16, lr=LR)
    batch = tu.make_batch(4, 64, 48)
    _, grads, _, _ = tu.run_step(shell, net, batch, DEV)
    whole = torch.cat([grads[n].reshape(-1) for n, _ in net.named_parameters()]).cpu()
    avg = ret[0][1]
    # app_gen (L1 mean) and the regulariser (mean over positions) average exactly over equal shards; the correctness
    # term is a mean over (b, positions) too -> the averaged shard gradients are the whole-batch gradient
    assert (avg - whole).abs().max().item() <= 1e-4 * whole.abs().max().item()
