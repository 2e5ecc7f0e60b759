"""Pins the CPU oracle (oracle/gfla_oracle.c): reference KATs, independent identities, autograd.

None of this touches the product library; it establishes that the checker used by the GPU parity
tests restates the reference kernels correctly.
"""
import pytest
import torch
import torch.nn.functional as F

from util import FLOW_KINDS, make_flow, max_abs, rand, randn


# ---- the reference's own known-answer tests ------------------------------------------------
def test_reshape_kat_from_reference_test(oracle):
    # test_local_attn_reshape.py:29-43: channels hold 0..8, k=3 -> top-left 3x3 block is 0..8 row-major
    x = torch.arange(9.0).view(1, 9, 1, 1).expand(4, 9, 14, 10).contiguous()
    out = oracle.local_attn_reshape_fwd(x, 3)
    assert out.shape == (4, 1, 42, 30)
    assert torch.equal(out[0, 0, :3, :3], torch.tensor([[0., 1., 2.], [3., 4., 5.], [6., 7., 8.]]))


def test_block_extractor_zero_flow_identity_from_reference_test(oracle):
    # test_block_extractor.py:46-49,55: zero flow => out[b,c,3:6,3:6] == source[b,c,0:3,0:3] (k=3)
    s = randn((4, 6, 14, 10), torch.float64, seed=1)
    z = torch.zeros(4, 2, 14, 10, dtype=torch.float64)
    out = oracle.block_extractor_fwd(s, z, 3)
    assert torch.equal(out[:, :, 3:6, 3:6], s[:, :, 0:3, 0:3])
    # centre tap is the identity, border replicates
    assert torch.equal(out[:, :, 1::3, 1::3], s)
    assert torch.equal(out[:, :, 0, 0], s[:, :, 0, 0])


# ---- independent identities (not derived from the reference) --------------------------------
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_reshape_is_pixel_shuffle(oracle, k, dtype):
    x = randn((3, k * k, 7, 5), dtype, seed=k)
    out = oracle.local_attn_reshape_fwd(x, k)
    assert torch.equal(out, F.pixel_shuffle(x, k))
    # backward is the inverse permutation
    g = randn(tuple(out.shape), dtype, seed=k + 10)
    assert torch.equal(oracle.local_attn_reshape_bwd(g, k), F.pixel_unshuffle(g, k))


@pytest.mark.parametrize("kind", FLOW_KINDS)
@pytest.mark.parametrize("k", [2, 3, 4, 5])
def test_block_extractor_fwd_identities(oracle, kind, k):
    s = randn((2, 5, 14, 10), torch.float64, seed=3)
    f = make_flow(kind, 2, 14, 10, torch.float64, seed=4)
    out = oracle.block_extractor_fwd(s, f, k)
    assert max_abs(out, oracle.block_extractor_gather(s, f, k)) < 1e-13
    assert max_abs(out, oracle.block_extractor_grid_sample(s, f, k)) < 1e-12
    if kind == "zero":  # replicate-padded unfold
        lo, hi = k // 2, k - 1 - k // 2
        sp = F.pad(s, (lo, hi, lo, hi), mode="replicate")
        unf = F.unfold(sp, k).view(2, 5, k, k, 14, 10).permute(0, 1, 4, 2, 5, 3).reshape(2, 5, 14 * k, 10 * k)
        assert torch.equal(out, unf)


def test_block_extractor_source_and_flow_sizes_may_differ(oracle):
    # external_function.py:62-66: grid (B,1,H,W) sampled with a (B,2,H-k+1,W-k+1) flow
    s = randn((2, 1, 12, 9), torch.float64, seed=5)
    f = torch.zeros(2, 2, 10, 7, dtype=torch.float64) + 1.0
    out = oracle.block_extractor_fwd(s, f, 3)
    assert out.shape == (2, 1, 30, 21)
    assert max_abs(out, oracle.block_extractor_gather(s, f, 3)) == 0.0


def test_block_extractor_f32_matches_f64_to_rounding(oracle):
    s, f = randn((2, 4, 16, 12), seed=6), make_flow("wild", 2, 16, 12, seed=7)
    o32 = oracle.block_extractor_fwd(s, f, 5)
    o64 = oracle.block_extractor_fwd(s.double(), f.double(), 5)
    assert max_abs(o32, o64) < 5e-6


@pytest.mark.parametrize("k", [3, 5])
def test_block_extractor_bwd_is_the_gradient(oracle, k):
    # the reference's gradcheck shapes: test_block_extractor.py:74-78
    s = rand((4, 6, 14, 10), torch.float64, seed=8).requires_grad_()
    f = (rand((4, 2, 14, 10), torch.float64, seed=9) * 1.8).requires_grad_()
    out = oracle.block_extractor_gather(s, f, k)
    g = randn(tuple(out.shape), torch.float64, seed=10)
    out.backward(g)
    gs, gf = oracle.block_extractor_bwd(s.detach(), f.detach(), g, k)
    assert max_abs(gs, s.grad) < 1e-12
    assert max_abs(gf, f.grad) < 1e-12


@pytest.mark.parametrize("k,d", [(2, 1), (4, 1), (4, 2), (6, 1)])
def test_resample2d_fwd_identity(oracle, k, d):
    i1 = randn((2, 4, 9, 8), torch.float64, seed=11)
    i2 = torch.cat((make_flow("wild", 2, 9, 8, torch.float64, seed=12),
                    rand((2, 1, 9, 8), torch.float64, seed=13) + 0.5), 1).contiguous()
    assert max_abs(oracle.resample2d_fwd(i1, i2, k, d), oracle.resample2d_gather(i1, i2, k, d)) < 1e-13


def test_resample2d_bwd_floor_variant_is_the_gradient_and_trunc_deviates(oracle):
    i1 = randn((2, 4, 9, 8), torch.float64, seed=14).requires_grad_()
    i2 = torch.cat((make_flow("coherent", 2, 9, 8, torch.float64, seed=15),
                    rand((2, 1, 9, 8), torch.float64, seed=16) + 1.0), 1).contiguous().requires_grad_()
    out = oracle.resample2d_gather(i1, i2, 4, 1)
    g = randn(tuple(out.shape), torch.float64, seed=17)
    out.backward(g)
    g1, g2 = oracle.resample2d_bwd(i1.detach(), i2.detach(), g, 4, 1, trunc_compat=False)
    assert max_abs(g1, i1.grad) < 1e-12
    assert max_abs(g2, i2.grad) < 1e-12   # dx, dy AND sigma
    # the reference's int() truncation (resample2d_kernel.cu:137-138) differs where x+dx<0 or y+dy<0
    g1t, g2t = oracle.resample2d_bwd(i1.detach(), i2.detach(), g, 4, 1, trunc_compat=True)
    assert max_abs(g2t, g2) == 0.0
    assert max_abs(g1t, g1) > 1e-3
    neg = ((torch.arange(8.).view(1, 1, 8) + i2[:, 0].detach()) < 0) | ((torch.arange(9.).view(1, 9, 1) + i2[:, 1].detach()) < 0)
    assert neg.any()


def test_resample2d_sigma_zero_and_module_sigma(oracle):
    i1 = randn((1, 2, 6, 6), torch.float32, seed=18)
    fl = make_flow("coherent", 1, 6, 6, seed=19)
    out = oracle.resample2d_module_fwd(i1, fl, 4, 1, 2.0)
    assert torch.isfinite(out).all()
    # sigma == 0 goes through SAFE_DIV's EPS arm (resample2d_kernel.cu:15); all weights underflow
    # except exact hits, the result must stay finite
    i2 = torch.cat((fl, torch.zeros(1, 1, 6, 6)), 1).contiguous()
    assert torch.isfinite(oracle.resample2d_fwd(i1, i2, 4, 1)).all()


@pytest.mark.parametrize("k", [3, 5])
def test_extractor_attn_aggregate_identity(oracle, k):
    # avg_pool(pixel_shuffle(a) * bs, k) == (1/k^2) sum_ij a_ij bs_ij  (what the fused kernel computes)
    s = randn((2, 6, 10, 8), torch.float64, seed=20)
    f = make_flow("coherent", 2, 10, 8, torch.float64, seed=21)
    a = torch.softmax(randn((2, k * k, 10, 8), torch.float64, seed=22), 1)
    bs = oracle.block_extractor_fwd(s, f, k)
    ref = F.avg_pool2d(oracle.local_attn_reshape_fwd(a.contiguous(), k) * bs, k, k)
    direct = torch.zeros_like(ref)
    for i in range(k):
        for j in range(k):
            direct += a[:, i * k + j].unsqueeze(1) * bs[:, :, i::k, j::k]
    assert max_abs(ref, direct / (k * k)) < 1e-14


def test_fc_split_identity(oracle):
    # conv0(cat(block_target, block_source)) == conv_s1(replicate_pad(target), W[:, :C]) + conv_sk(block_source, W[:, C:])
    k, C = 3, 4
    s, t = randn((2, C, 9, 7), torch.float64, seed=23), randn((2, C, 9, 7), torch.float64, seed=24)
    f = make_flow("coherent", 2, 9, 7, torch.float64, seed=25)
    w, b = randn((8, 2 * C, k, k), torch.float64, seed=26), randn((8,), torch.float64, seed=27)
    bs = oracle.block_extractor_fwd(s, f, k)
    bt = oracle.block_extractor_fwd(t, torch.zeros_like(f), k)
    full = F.conv2d(torch.cat((bt, bs), 1), w, b, stride=k)
    lo, hi = k // 2, k - 1 - k // 2
    split = F.conv2d(F.pad(t, (lo, hi, lo, hi), mode="replicate"), w[:, :C], b) + F.conv2d(bs, w[:, C:], None, stride=k)
    assert max_abs(full, split) < 1e-12


@pytest.mark.parametrize("kz", [3, 5])
def test_affine_regularization_collapses_to_a_quadratic_form(oracle, kz):
    # external_function.py:61-69 op by op (oracle kernels) == mean(u^T M u) (losses.py)
    from global_flow_local_attention_amd.losses import AffineRegularizationLoss, affine_projector
    flow = make_flow("coherent", 2, 12, 10, torch.float64, seed=40)
    loss_mod = AffineRegularizationLoss(kz)            # collapsed: pure torch, runs on the CPU
    got = loss_mod(flow)
    grid = loss_mod.flow2grid(flow)
    weights = affine_projector(kz).view(kz * kz, 1, kz, kz)
    want = 0
    for ax in (0, 1):
        g = grid[:, ax:ax + 1].contiguous()
        results = F.conv2d(g, weights)
        b, c, h, w = results.shape
        kernels_new = oracle.local_attn_reshape_fwd(results.contiguous(), kz)
        f = torch.zeros(b, 2, h, w, dtype=torch.float64) + float(int(kz / 2))
        grid_h = oracle.block_extractor_fwd(g, f, kz)
        want = want + F.avg_pool2d(grid_h * kernels_new, kz, kz).mean() * kz ** 2
    assert abs(got.item() - want.item()) <= 1e-10 * max(1.0, abs(want.item()))


def test_affine_regularization_loss_reference_golden(oracle):
    """tests/golden/affine_golden.npz holds what the REFERENCE's own AffineRegularizationLoss class computed
    (external_function.py:31-77, run on the host by tests/golden/make_affine_golden.py): value and d/d flow.  Both the
    collapsed quadratic form (losses.py, the product) and the op-by-op restatement (oracle/cpu_modules.py) must match."""
    import os
    import numpy as np
    from global_flow_local_attention_amd.losses import AffineRegularizationLoss
    from oracle.cpu_modules import AffineRegularizationLossOpByOp
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "affine_golden.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    assert len(names) == 3
    for name in names:
        kz = int(name[2])
        want, want_g = float(z[name + "/loss"]), torch.from_numpy(z[name + "/g_flow"])
        for mod in (AffineRegularizationLoss(kz), AffineRegularizationLossOpByOp(kz)):
            f = torch.from_numpy(z[name + "/flow"]).clone().requires_grad_()
            loss = mod(f)
            loss.backward()
            assert abs(loss.item() - want) <= 2e-5 * max(1.0, abs(want)), (name, type(mod).__name__, loss.item(), want)
            assert (f.grad - want_g).abs().max().item() <= 1e-4 * max(1.0, want_g.abs().max().item())
