"""The face model's pair of ExtractorAttn per attention layer (generator.py:490-499) on two HIP streams
(global_flow_local_attention_amd/face_step.py): same results as the sequential evaluation on one stream, forward and
every gradient, f32 and bf16 features, and as the host oracle's op-by-op blocks."""
import pytest
import torch

from util import make_flow, rand, randn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(B, C, H, W, dtype, seed):
    t = lambda x: x.to(dtype).to(DEV).requires_grad_()
    out, prev, ref = (t(randn((B, C, H, W), seed=seed + i)) for i in range(3))
    fp, fr = t(make_flow("smooth", B, H, W, seed=seed + 3)), t(make_flow("coherent", B, H, W, seed=seed + 4))
    mp, mr = rand((B, 1, H, W), seed=seed + 5).to(dtype).to(DEV), rand((B, 1, H, W), seed=seed + 6).to(dtype).to(DEV)
    return out, prev, ref, fp, fr, mp, mr


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,H,W,k", [(32, 16, 12, 3), (16, 24, 20, 5)])
def test_dual_stream_pair_equals_sequential(gfla, dtype, C, H, W, k):
    B = 3
    torch.manual_seed(0)
    attn_p = gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True).to(DEV)
    attn_r = gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True).to(DEV)
    up = randn((B, C, H, W), seed=50).to(dtype).to(DEV)
    results = []
    for dual in (True, False):
        pair = gfla.DualStreamAttn(attn_p, attn_r, enabled=dual)
        args = _inputs(B, C, H, W, dtype, seed=10)
        for p in list(attn_p.parameters()) + list(attn_r.parameters()):
            p.grad = None
        for _ in range(3):   # repeated calls reuse the side stream; the last one is compared
            res = pair(*args)
        for a in args[:5]:
            a.grad = None
        for p in list(attn_p.parameters()) + list(attn_r.parameters()):
            p.grad = None
        res = pair(*args)
        res.backward(up)
        torch.cuda.synchronize()
        grads = [a.grad.float() for a in args[:5]] + [p.grad.float() for p in list(attn_p.parameters()) + list(attn_r.parameters())]
        results.append((res.detach().float(), grads))
    (r2, g2), (r1, g1) = results
    assert torch.equal(r2, r1)                       # the forward has no atomics: identical bits on either schedule
    tol = 1e-5 if dtype == torch.float32 else 2 ** -7
    for a, b in zip(g2, g1):
        assert (a - b).abs().max().item() <= tol * max(1e-30, b.abs().max().item())


def test_dual_stream_pair_against_host_oracle_blocks(gfla, oracle):
    from oracle import cpu_modules
    B, C, H, W, k = 2, 16, 14, 10, 3
    torch.manual_seed(1)
    mods = [gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True) for _ in range(2)]
    refs = []
    for m in mods:
        r = cpu_modules.ExtractorAttnCPU(C, k, torch.nn.LeakyReLU(0.1), softmax=True)
        r.load_state_dict(m.state_dict())
        refs.append(r)
        m.to(DEV)
    args = _inputs(B, C, H, W, torch.float32, seed=20)
    res = gfla.DualStreamAttn(mods[0], mods[1])(*args)
    res.sum().backward()
    out, prev, ref, fp, fr, mp, mr = [a.detach().cpu().clone().requires_grad_(a.requires_grad) for a in args]
    want = out * (1 - mp) + refs[0](prev, out, fp) * mp + out * (1 - mr) + refs[1](ref, out, fr) * mr   # generator.py:494-499
    want.sum().backward()
    assert (res.detach().cpu() - want.detach()).abs().max().item() <= 2e-6
    for got, w in zip(args[:5], (out, prev, ref, fp, fr)):
        assert (got.grad.cpu() - w.grad).abs().max().item() <= 1e-5 * max(1.0, w.grad.abs().max().item())


def test_generate_frames_recurrence(gfla):
    """generate_frames mirrors FaceGenerator.forward's recurrence (generator.py:406-426): frame t's image is frame t+1's
    `previous`; the first frame's previous is the reference when none is given."""
    seen = []

    def frame_fn(t, previous, reference):
        seen.append((t, previous, reference))
        return "img%d" % t
    images = gfla.generate_frames(frame_fn, 3, None, "ref")
    assert images == ["img0", "img1", "img2"]
    assert seen == [(0, "ref", "ref"), (1, "img0", "ref"), (2, "img1", "ref")]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 20, 9, 7), (3, 16, 16, 12), (1, 130, 5, 33)])
def test_mask_blend_kernel_equals_op_by_op(gfla, dtype, shape):
    """csrc/mask_blend.hip against the reference's expression (generator.py:496-499) evaluated op by op in torch: forward
    bit for bit (f32 and bf16: every intermediate is rounded where torch rounds it), all five gradients incl. the masks'
    (sums over the channels)."""
    B, C, H, W = shape
    mk = lambda s, seed: randn(s, seed=seed).to(dtype).to(DEV).requires_grad_()
    out, a_p, a_r = mk(shape, 1), mk(shape, 2), mk(shape, 3)
    m_p = rand((B, 1, H, W), seed=4).to(dtype).to(DEV).requires_grad_()
    m_r = rand((B, 1, H, W), seed=5).to(dtype).to(DEV).requires_grad_()
    up = randn(shape, seed=6).to(dtype).to(DEV)
    leaves = (out, a_p, a_r, m_p, m_r)
    y = gfla.MaskBlendFunction.apply(*leaves)
    y.backward(up)
    got = [t.grad.float().clone() for t in leaves]
    for t in leaves:
        t.grad = None
    want = (out * (1 - m_p) + a_p * m_p) + (out * (1 - m_r) + a_r * m_r)
    want.backward(up)
    assert torch.equal(y, want)
    tol = 1e-5 if dtype == torch.float32 else 2 ** -6
    for g, t, name in zip(got, leaves, ("out", "attn_p", "attn_r", "mask_p", "mask_r")):
        w = t.grad.float()
        assert (g - w).abs().max().item() <= tol * max(1e-30, w.abs().max().item()), name


@pytest.mark.parametrize("sizes", [(1,), (7, 4096), (33, 1000003, 5), (2049, 16, 777, 8193), (5, 6, 7, 8, 9, 10)])
def test_convert_many_equals_torch(gfla, sizes):
    """gfla_convert_multi (one launch for up to four tensors, the bf16 feature path's conversions): bit-identical to
    torch's .to() both ways -- float32 -> bfloat16 rounds to nearest even, NaN / inf / denormals included; odd sizes and
    unaligned views take the scalar tail."""
    from global_flow_local_attention_amd import _lib
    gen = torch.Generator(device=DEV).manual_seed(len(sizes))
    src32 = []
    for i, n in enumerate(sizes):
        t = torch.randn(n + 1, device=DEV, generator=gen) * (10.0 ** (i - 2))
        if n >= 5:
            t[1], t[2], t[3], t[4] = float("nan"), float("inf"), -float("inf"), 1e-40
        src32.append(t[1:] if i % 2 else t[:n])   # every other tensor starts 4 bytes off a 16-byte boundary
    want16 = [t.to(torch.bfloat16) for t in src32]
    got16 = _lib.convert_many(src32 + [None], torch.bfloat16)
    assert got16[-1] is None
    for g, w in zip(got16, want16):
        assert g.dtype == torch.bfloat16 and torch.equal(g.view(torch.int16), w.view(torch.int16))
    back = _lib.convert_many(want16, torch.float32)
    for g, w in zip(back, want16):
        assert g.dtype == torch.float32 and torch.equal(g.view(torch.int32), w.float().view(torch.int32))
    same = _lib.convert_many(src32[:1], torch.float32)   # nothing to convert: handed back unchanged
    assert same[0] is src32[0] or torch.equal(same[0], src32[0])
