"""Sampling-correctness loss on the GPU: the MFMA best-match kernel and the loss built on it, against the
host restatement and the golden vectors of the reference's own calculate_loss."""
import numpy as np
import pytest
import torch

from oracle.cpu_modules import PerceptualCorrectnessCPU, max_cosine_cpu
from test_correctness_cpu import CASES, PATH, case
from util import assert_close, make_flow, randn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def features(shape, seed):
    """post-ReLU-like features: non-negative, a few exact zeros, different norms per position"""
    x = randn(shape, seed=seed).relu() * (1 + randn(shape[:1] + (1,) + shape[2:], seed=seed + 1).abs())
    return x.contiguous()


def check_best(best, index, src, tgt, tol=2e-6):
    """value against the host bmm/max; index by the value it points at (ties may resolve differently)"""
    want, _ = max_cosine_cpu(src.double(), tgt.double())
    assert_close(best.cpu(), want, tol, "best")
    s = src.double() / (src.double().norm(dim=1, keepdim=True) + 1e-8)
    t = tgt.double() / (tgt.double().norm(dim=1, keepdim=True) + 1e-8)
    picked = torch.gather(s, 2, index.cpu().long().unsqueeze(1).expand(-1, s.size(1), -1))
    assert_close((picked * t).sum(1), want, tol, "value at index")
    assert int(index.min()) >= 0 and int(index.max()) < src.size(2)


@pytest.mark.parametrize("B,C,Ns,Nt", [
    (2, 16, 120, 120),      # one ragged tile
    (1, 64, 128, 128),      # exactly one tile, 4 channel chunks
    (3, 20, 300, 257),      # C not a multiple of the chunk, Nt odd (scalar staging), ragged both ways
    (2, 7, 130, 5),         # fewer channels than one chunk, a handful of targets
    (1, 33, 1, 200),        # a single source position
    (2, 256, 704, 704),     # relu3_1-like channel count, 32x22 positions
    (1, 48, 1023, 515),     # Ns odd
])
def test_max_cosine_matches_host_bmm_max(gfla, B, C, Ns, Nt):
    src, tgt = features((B, C, Ns), 1), features((B, C, Nt), 2)
    best, index = gfla.max_cosine_similarity(src.to(DEV), tgt.to(DEV), return_index=True)
    assert best.shape == (B, Nt) and index.dtype == torch.int32
    check_best(best, index, src, tgt)


@pytest.mark.parametrize("split", [1, 2, 3, 64])
def test_max_cosine_source_range_split(gfla, split):
    """units over ranges of source tiles merge through the packed atomic max (tuning key 5 forces the split)"""
    src, tgt = features((2, 32, 1000), 11), features((2, 32, 260), 12)
    old = gfla.set_tuning(5, split)
    try:
        best, index = gfla.max_cosine_similarity(src.to(DEV), tgt.to(DEV), return_index=True)
    finally:
        gfla.set_tuning(5, old)
    check_best(best, index, src, tgt)


def test_max_cosine_mixed_sign_and_zero_vectors(gfla):
    src, tgt = randn((2, 24, 200), seed=3), randn((2, 24, 150), seed=4)
    src[:, :, 7] = 0          # zero vectors: 0/(0+eps) = 0 similarity, as in the reference
    tgt[:, :, 11] = 0
    best, index = gfla.max_cosine_similarity(src.to(DEV), tgt.to(DEV), return_index=True)
    check_best(best, index, src, tgt)
    assert best[:, 11].abs().max().item() == 0.0
    # all similarities negative: the maximum must not be the zero of a padded row
    neg_src = -features((1, 16, 130), 5) - 0.1
    pos_tgt = features((1, 16, 40), 6) + 0.1
    best = gfla.max_cosine_similarity(neg_src.to(DEV), pos_tgt.to(DEV))
    assert best.max().item() < 0
    check_best(best, gfla.max_cosine_similarity(neg_src.to(DEV), pos_tgt.to(DEV), return_index=True)[1], neg_src, pos_tgt)


def test_max_cosine_full_size_properties(gfla):
    """BASELINE shapes (B=32/GPU, relu3_1 (256, 64x44), relu4_1 (512, 32x22)): the host bmm would need
    minutes, so use what must hold at any size: every position matches itself when target is a permutation
    of source (best = 1, index = the permutation)."""
    for B, C, N in ((32, 256, 64 * 44), (32, 512, 32 * 22)):
        g = torch.Generator(device=DEV).manual_seed(N)
        src = torch.randn(B, C, N, device=DEV, generator=g)
        perm = torch.randperm(N, device=DEV, generator=g)
        best, index = gfla.max_cosine_similarity(src, src[:, :, perm].contiguous(), return_index=True)
        assert (best - 1).abs().max().item() <= 2e-6
        assert torch.equal(index.long(), perm.unsqueeze(0).expand(B, -1))
        # and a slice of the real thing against the host
        tgt = torch.randn(B, C, N, device=DEV, generator=g)
        best = gfla.max_cosine_similarity(src, tgt)
        want, _ = max_cosine_cpu(src[:2].cpu().double(), tgt[:2, :, :256].cpu().double())
        assert_close(best[:2, :256].cpu(), want, 2e-6, "slice")


def test_max_cosine_gradients(gfla):
    src = features((2, 12, 90), 7).requires_grad_()
    tgt = features((2, 12, 70), 8).requires_grad_()
    up = randn((2, 70), seed=9)
    want, _ = max_cosine_cpu(src, tgt)
    (want * up).sum().backward()
    s, t = src.detach().to(DEV).requires_grad_(), tgt.detach().to(DEV).requires_grad_()
    (gfla.max_cosine_similarity(s, t) * up.to(DEV)).sum().backward()
    assert_close(s.grad.cpu(), src.grad, 1e-5, "grad source")
    assert_close(t.grad.cpu(), tgt.grad, 1e-5, "grad target")


@pytest.mark.parametrize("B,C,N", [(2, 12, 70), (1, 64, 64), (3, 7, 129), (2, 256, 704)])
def test_correctness_map_fused_vs_torch(gfla, B, C, N):
    """exp(-cosine_similarity(x, t) / (best + eps)) and its three gradients against torch in fp64 on the host,
    including a zero vector and one shorter than cosine_similarity's eps"""
    x, t = features((B, C, N), 21), features((B, C, N), 22)
    x[0, :, 3] = 0
    t[0, :, 5] = 0
    x[0, :, 7] *= 1e-10
    best = (randn((B, N), seed=23).abs() * 0.5 + 0.2).contiguous()
    up = randn((B, N), seed=24)
    ref = [v.double().requires_grad_() for v in (x, t, best)]
    want = torch.exp(-torch.nn.functional.cosine_similarity(ref[0], ref[1]) / (ref[2] + 1e-8))
    (want * up.double()).sum().backward()
    dev = [v.to(DEV).requires_grad_() for v in (x, t, best)]
    got = gfla.CorrectnessMapFunction.apply(dev[0], dev[1], dev[2], 1e-8)
    (got * up.to(DEV)).sum().backward()
    assert_close(got.detach().cpu(), want.detach(), 2e-6, "loss map")
    for d, r, what in zip(dev, ref, ("grad warped", "grad target", "grad best")):
        ok = torch.ones_like(r.grad, dtype=torch.bool)
        if what != "grad best":
            ok[0, :, 7] = False     # |x| = 1e-10: fp32 underflows the squared norm; checked separately below
        assert_close(d.grad.cpu()[ok], r.grad[ok], 1e-5, what)
    assert torch.isfinite(dev[0].grad).all() and torch.isfinite(dev[1].grad).all()
    # only some gradients requested
    only = [x.to(DEV).requires_grad_(), t.to(DEV), best.to(DEV)]
    gfla.CorrectnessMapFunction.apply(*only, 1e-8).sum().backward()
    assert only[0].grad is not None and only[1].grad is None


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", CASES)
def test_correctness_loss_vs_reference_golden(gfla, name, fused):
    g = case(np.load(PATH), name, DEV)
    src, tgt, flow = (g[k].clone().requires_grad_() for k in ("src", "tgt", "flow"))
    mod = gfla.PerceptualCorrectness()
    mod.fused = fused
    mod.target_vgg, mod.source_vgg = {"f": tgt}, {"f": src}
    best, index = gfla.max_cosine_similarity(src, tgt, return_index=True)
    assert_close(best.detach(), g["best"], 2e-6, "best")
    loss = mod.calculate_loss(flow, "f", g["mask"])
    loss.backward()
    assert abs(loss.item() - g["loss"].item()) <= 2e-6
    assert_close(flow.grad, g["g_flow"], 1e-5, "grad flow")
    assert_close(src.grad, g["g_src"], 1e-5, "grad source features")
    assert_close(tgt.grad, g["g_tgt"], 1e-5, "grad target features")


def test_correctness_loss_call_with_injected_features_and_bilinear_variant(gfla):
    B, H, W = 2, 24, 20
    feats = lambda img: {"relu3_1": torch.nn.functional.avg_pool2d(img, 2).repeat(1, 8, 1, 1).contiguous(),
                         "relu4_1": torch.nn.functional.avg_pool2d(img, 4).repeat(1, 12, 1, 1).contiguous()}
    tgt_img, src_img = randn((B, 3, H, W), seed=1).abs(), randn((B, 3, H, W), seed=2).abs()
    flows = [make_flow("coherent", B, H // 4, W // 4, seed=3), make_flow("smooth", B, H // 2, W // 2, seed=4)]
    mod = gfla.PerceptualCorrectness(vgg=feats)
    got = mod(tgt_img.to(DEV), src_img.to(DEV), [f.to(DEV) for f in flows], [2, 3])
    ref = PerceptualCorrectnessCPU()
    ref.target_vgg, ref.source_vgg = feats(tgt_img), feats(src_img)
    want = ref.calculate_loss(flows[0], "relu4_1") + ref.calculate_loss(flows[1], "relu3_1")
    assert abs(got.item() - want.item()) <= 2e-6
    warped = mod.bilinear_warp(feats(src_img)["relu3_1"].to(DEV), flows[1].to(DEV))
    assert warped.shape == (B, 24, (H // 2) * (W // 2))
    with pytest.raises(RuntimeError):
        gfla.PerceptualCorrectness()(tgt_img, src_img, flows, [2, 3])
