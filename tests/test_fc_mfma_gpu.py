"""ExtractorAttn's fully_connect_layer on the matrix cores (csrc/fc_gemm.hip, fc_sample.hip, fc_block.hip) against
float64 restatements of base_function.py:799-807, against the CPU oracle of the whole block, against the real
reference kernels (oracle/_ref, when built) and against the library-GEMM path of round 1.

Tolerances (max abs error relative to the largest reference entry): the three arithmetic modes are held to the SAME
bars -- 1e-5 for the convolved maps / logits, 2e-5 for gradients -- so a split-precision mode cannot pass where the
exact-f32 mode would not.  The measured errors against float64 are printed (pytest -s) and recorded in DESIGN.md.
"""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

from util import assert_close, make_flow, max_abs, randn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODES = (0, 4, 5, 3, 2)
FWD_TOL, GRAD_TOL = 1e-5, 2e-5


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


@pytest.fixture(scope="module")
def lib(gfla):
    from global_flow_local_attention_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    return _lib


def rel_err(got, want):
    return max_abs(got, want) / max(1e-30, want.double().abs().max().item())


# ------------------------------------------------------------------------------- the transposing LDS read
def test_tr_read_semantics(lib):
    """ds_read_b64_tr_b16 as the weight-gradient kernel assumes it: inside each group of 16 lanes, lane i / element j
    receives element (i & 3) of the 8 bytes addressed by lane 4*j + (i >> 2)."""
    n = 4096
    image = torch.arange(n, dtype=torch.int16, device=DEV)
    lanes = torch.arange(64)
    for name, elem_off in (("linear", lanes * 4),
                           ("rows of 40", (lanes >> 2) * 40 + (lanes & 3) * 4),
                           ("wgrad X", ((lanes & 15) >> 2) * 16 + (lanes >> 5) * 8 * 16 + ((lanes >> 4) & 1) * 3 * 16
                            + (lanes & 3) * 4)):
        off = (elem_off * 2).to(torch.int32).to(DEV)
        out = torch.zeros(64, 4, dtype=torch.int16, device=DEV)
        lib.call("gfla_fc_tr_probe", image, _ptr(image), n, _ptr(off), _ptr(out))
        got = out.cpu().long()
        want = torch.zeros(64, 4, dtype=torch.long)
        for lane in range(64):
            i = lane & 15
            for j in range(4):
                supplier = (lane & 48) + 4 * j + (i >> 2)
                want[lane, j] = int(elem_off[supplier]) + (i & 3)
        assert torch.equal(got, want), "%s: got\n%s\nwant\n%s" % (name, got[:20], want[:20])


# ------------------------------------------------------------------------------- the convolutions in isolation
def _half_weights(w0, C, is_source):
    return w0[:, C:] if is_source else w0[:, :C]


def _pads(k, is_source):
    lo, hi = k // 2, k - 1 - k // 2
    return (k - 1, k - 1, k - 1, k - 1) if is_source else (lo, hi, lo, hi)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k,C,H,W", [(3, 6, 10, 8), (5, 20, 9, 13), (3, 32, 32, 22), (5, 16, 24, 20)])
@pytest.mark.parametrize("is_source", [0, 1])
def test_conv_fwd_bwd_one_half(lib, gfla, mode, k, C, H, W, is_source):
    from global_flow_local_attention_amd import fc_mfma
    B = 2
    x = (randn((B, C, H, W), seed=1) * 1.7).to(DEV)
    w0 = (randn((128, 2 * C, k, k), seed=2) * 0.05).to(DEV)
    g = fc_mfma.geometry(H, W, k, is_source)
    ws = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 0), dtype=torch.uint8, device=DEV)
    out = torch.full((B, g["Mg"], 128), float("nan"), device=DEV)
    lib.call("gfla_fc_conv_fwd_f32", x, _ptr(x), _ptr(w0), is_source, _ptr(ws), _ptr(out), B, C, H, W, k, mode)
    x64 = x.cpu().double().requires_grad_()                                         # float64 reference on the host
    wh = _half_weights(w0, C, is_source).cpu().double().clone().requires_grad_()
    ref = F.conv2d(F.pad(x64, _pads(k, is_source), mode="replicate"), wh)           # (B,128,Ho,Wo)
    assert ref.shape[2:] == (g["Ho"], g["Wo"])
    rows = (torch.arange(g["Ho"])[:, None] * g["Wp"] + torch.arange(g["Wo"])[None, :]).reshape(-1).to(DEV)
    got = out[:, :g["Ho"] * g["Wo"], :].reshape(B, g["Ho"], g["Wo"], 128).permute(0, 3, 1, 2)  # compact rows yo*Wo + xo
    e_f = rel_err(got.cpu(), ref.detach().cpu())
    assert e_f <= FWD_TOL, "convolved map, mode %d: rel err %.3e" % (mode, e_f)

    # backward from a gradient map in Z layout
    dG = randn((B, 128, g["Ho"], g["Wo"]), seed=3).to(DEV) * 1e-3
    z = torch.zeros(B, g["Sz"], 128, device=DEV)
    z[:, g["lead"] + rows, :] = dG.permute(0, 2, 3, 1).reshape(B, -1, 128)
    ref.backward(dG.cpu().double())
    scratch = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 1), dtype=torch.uint8, device=DEV)
    gx = torch.full((B, C, H, W), float("nan"), device=DEV)
    gw = torch.full((128, 2 * C, k, k), float("nan"), device=DEV)
    lib.call("gfla_fc_conv_bwd_f32", x, _ptr(z), is_source, _ptr(ws), _ptr(scratch), _ptr(gx), _ptr(gw), B, C, H, W, k,
             mode)
    e_x = rel_err(gx.cpu(), x64.grad.cpu())
    e_w = rel_err(_half_weights(gw, C, is_source).cpu(), wh.grad.cpu())
    other = _half_weights(gw, C, 1 - is_source)
    print("mode %d k %d C %d %dx%d half %d: rel err map %.2e grad_x %.2e grad_w %.2e" % (mode, k, C, H, W, is_source,
                                                                                     e_f, e_x, e_w))
    assert e_x <= GRAD_TOL, "data gradient, mode %d: rel err %.3e" % (mode, e_x)
    assert e_w <= GRAD_TOL, "weight gradient, mode %d: rel err %.3e" % (mode, e_w)
    assert float(other.abs().max()) == 0.0


# ------------------------------------------------------------------------------- the whole layer
def _logits_f64(s, t, f, w0, b0, w1, b1, k, slope):
    """base_function.py:805-807 + fully_connect_layer[:3] in float64 on the host: the oracle's literal
    block_extractor kernels (oracle/cpu_modules.py) + torch convolutions."""
    from oracle import cpu_modules
    bs = cpu_modules._BlockExtractorCPU.apply(s, f, k)
    bt = cpu_modules._BlockExtractorCPU.apply(t, torch.zeros_like(f), k)
    hidden = F.conv2d(torch.cat((bt, bs), 1), w0, b0, stride=k)
    return F.conv2d(F.leaky_relu(hidden, slope), w1.reshape(k * k, 128, 1, 1), b1), hidden.detach()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k,C,H,W,kind", [(3, 16, 10, 8, "coherent"), (5, 8, 12, 9, "wild"), (3, 24, 16, 11, "smooth"),
                                          (5, 6, 7, 10, "integer"), (3, 16, 9, 9, "zero")])
def test_fc_function_against_float64(lib, gfla, oracle, mode, k, C, H, W, kind):
    from global_flow_local_attention_amd.fc_mfma import FcMfmaFunction
    B, slope = 2, 0.1
    for seed in range(0, 200, 10):  # keep every hidden activation clear of the LeakyReLU kink (an f32-vs-f64 sign flip
        s, t = randn((B, C, H, W), seed=11 + seed).to(DEV), randn((B, C, H, W), seed=12 + seed).to(DEV)  # is not an error)
        f = make_flow(kind, B, H, W, seed=13 + seed).to(DEV)
        w0 = (randn((128, 2 * C, k, k), seed=14 + seed) / (2 * C * k * k) ** 0.5).to(DEV)
        b0 = (randn((128,), seed=15 + seed) * 0.1).to(DEV)
        w1 = (randn((k * k, 128, 1, 1), seed=16 + seed) / 128 ** 0.5).to(DEV)
        b1 = (randn((k * k,), seed=17 + seed) * 0.1).to(DEV)
        a64 = [x.cpu().double().clone().requires_grad_() for x in (s, t, f, w0, b0, w1, b1)]
        want, hidden = _logits_f64(*a64, k, slope)
        if hidden.abs().min().item() > 2e-5:
            break
    up = randn((B, k * k, H, W), seed=18).to(DEV)
    a32 = [x.clone().requires_grad_() for x in (s, t, f, w0, b0, w1, b1)]
    got = FcMfmaFunction.apply(*a32, k, slope, mode)
    e = rel_err(got.detach().cpu(), want.detach().cpu())
    assert e <= FWD_TOL, "logits, mode %d: %.3e" % (mode, e)
    got.backward(up)
    want.backward(up.cpu().double())
    names = ("source", "target", "flow", "w0", "b0", "w1", "b1")
    errs = []
    for n_, a, r in zip(names, a32, a64):
        if n_ == "flow" and kind in ("integer", "zero"):
            continue  # at integer positions the bilinear kink makes the one-sided derivative a convention
        errs.append((n_, rel_err(a.grad.cpu(), r.grad.cpu())))
    print("mode %d k %d %s: logits %.2e " % (mode, k, kind, e) + " ".join("%s %.2e" % x for x in errs))
    for n_, err in errs:
        assert err <= GRAD_TOL, "grad %s, mode %d: rel err %.3e" % (n_, mode, err)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k,C", [(3, 16), (5, 8)])
def test_extractor_attn_mfma_vs_library_path_and_oracle(lib, gfla, oracle, mode, k, C):
    """Module level: the MFMA path against the round-1 library-GEMM path, and forward + all gradients against the CPU
    oracle of the whole block (oracle/cpu_modules.py: the reference's op-by-op composition with the literal C
    restatement of the reference kernels)."""
    from oracle import cpu_modules
    B, H, W = 2, 12, 10
    for seed in range(20):  # keep the hidden activations clear of the LeakyReLU kink (see __graft_entry__.smoke)
        torch.manual_seed(seed)
        m = gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True)
        ref = cpu_modules.ExtractorAttnCPU(C, k, torch.nn.LeakyReLU(0.1), softmax=True)
        ref.load_state_dict(m.state_dict())
        s, t = randn((B, C, H, W), seed=41 + seed), randn((B, C, H, W), seed=42 + seed)
        f = make_flow("coherent", B, H, W, seed=43 + seed)
        hidden = []
        hook = ref.fully_connect_layer[0].register_forward_hook(lambda mod, i, o: hidden.append(o.detach()))
        with torch.no_grad():
            ref(s, t, f)
        hook.remove()
        if hidden[0].abs().min().item() > 1e-4:
            break
    m = m.to(DEV)
    up = randn((B, C, H, W), seed=44)

    def run(impl):
        m.fc_impl, m.fc_mode = impl, mode
        args = [x.to(DEV).requires_grad_() for x in (s, t, f)]
        m.zero_grad()
        attn, out = m.hook_attn_param(*args)
        out.backward(up.to(DEV))
        return out.detach().cpu(), attn.detach().cpu(), [a.grad.cpu() for a in args] + [p.grad.cpu() for p in m.parameters()]

    out_m, attn_m, g_m = run("mfma")
    out_l, attn_l, g_l = run("library")
    cargs = [x.clone().requires_grad_() for x in (s, t, f)]
    want = ref(*cargs)
    want.backward(up)
    g_c = [a.grad for a in cargs] + [p.grad for p in ref.parameters()]
    assert_close(out_m, out_l, 2e-5, "mfma vs library path, forward")
    assert_close(attn_m, attn_l, 2e-5, "mfma vs library path, attention")
    assert_close(out_m, want.detach(), 2e-5, "mfma path vs CPU oracle, forward")
    names = ["source", "target", "flow"] + [n for n, _ in m.named_parameters()]
    for n_, a, b_, c_ in zip(names, g_m, g_l, g_c):
        assert_close(a, b_, 1e-4, "grad %s: mfma vs library path" % n_)
        assert_close(a, c_, 1e-4, "grad %s: mfma path vs CPU oracle" % n_)


def test_unsupported_shapes_fall_back(gfla):
    """Other kernel sizes / dtypes keep running through the library path; mismatched target sizes through the
    reference's own composition."""
    m = gfla.ExtractorAttn(8, 4, torch.nn.LeakyReLU(0.1), softmax=True).to(DEV)
    s, t = randn((1, 8, 6, 6), seed=1).to(DEV), randn((1, 8, 6, 6), seed=2).to(DEV)
    f = make_flow("coherent", 1, 6, 6, seed=3).to(DEV)
    a = m(s, t, f)
    m.fused = False
    assert_close(a.cpu(), m(s, t, f).cpu(), 2e-5, "k=4 falls back")
    m3 = gfla.ExtractorAttn(8, 3, torch.nn.LeakyReLU(0.1), softmax=True).to(DEV)
    m3.fc_impl = "library"
    b = m3(s, t, f)
    m3.fc_impl = "mfma"
    assert_close(m3(s, t, f).cpu(), b.cpu(), 2e-5, "fc_impl switch")


def test_vendor_fallback_policy(gfla):
    """A configuration the library's own MFMA kernels do not take (the reference's constructor default kernel_size=4,
    base_function.py:791) reaches rocBLAS / MIOpen only when that is allowed: the package default warns once,
    install(strict_mfma=True) turns it into an error, a module attribute overrides either; every call that took
    the vendor path is counted (bench.py reports the count of its run)."""
    import warnings
    from global_flow_local_attention_amd import extractor_attn as ea
    m = gfla.ExtractorAttn(8, 4, torch.nn.LeakyReLU(0.1), softmax=True).to(DEV)
    s, t = randn((1, 8, 6, 6), seed=1).to(DEV), randn((1, 8, 6, 6), seed=2).to(DEV)
    f = make_flow("coherent", 1, 6, 6, seed=3).to(DEV)
    old = ea.VENDOR_FALLBACK
    try:
        ea.VENDOR_FALLBACK = "error"      # what install(strict_mfma=True) and bench.py set
        n0 = ea.vendor_fallback_calls
        with pytest.raises(ea.VendorFallbackError):
            m(s, t, f)
        m.vendor_fallback = "allow"       # per-module override
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            a = m(s, t, f)
        assert ea.vendor_fallback_calls == n0 + 2
        del m.vendor_fallback
        ea.VENDOR_FALLBACK = "warn"
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            b = m(s, t, f)
            m(s, t, f)
        assert sum("rocBLAS" in str(x.message) for x in w) == 1      # once per module
        assert torch.equal(a, b)
        # the supported configurations never count
        m3 = gfla.ExtractorAttn(8, 3, torch.nn.LeakyReLU(0.1), softmax=True).to(DEV)
        n1 = ea.vendor_fallback_calls
        ea.VENDOR_FALLBACK = "error"
        m3(s, t, f)
        assert ea.vendor_fallback_calls == n1
        m3.fc_impl = "library"            # an explicit request is always honoured (and counted)
        m3(s, t, f)
        assert ea.vendor_fallback_calls == n1 + 1
    finally:
        ea.VENDOR_FALLBACK = old


# ------------------------------------------------------------------------------- bf16 features (BASELINE config 5)
@pytest.mark.parametrize("k,C,H,W", [(3, 16, 12, 10), (5, 8, 11, 9)])
def test_extractor_attn_bf16_features(lib, gfla, oracle, k, C, H, W):
    """bf16 source / target / flow through ExtractorAttn (FusedAttnBf16Function: FC layers in arithmetic mode 1 --
    one f16 term per operand, exact for bf16 values -- aggregation in the _bf16 kernels) against the CPU oracle of the
    whole block evaluated in float32 on the bf16-rounded inputs: forward and feature-map gradients within 2^-6 of the largest
    entry, parameter gradients within 2^-5 (the logits are rounded to bf16 once before the softmax, as in any bf16 pipeline)."""
    from oracle import cpu_modules
    B, tol = 2, 2 ** -6
    torch.manual_seed(3)
    m = gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True)
    with torch.no_grad():   # keep the hidden activations clear of the LeakyReLU kink (module docstring)
        m.fully_connect_layer[0].bias.copy_(torch.where(torch.arange(128) % 2 == 0, 8.0, -8.0))
        for p in m.parameters():
            p.copy_(p.bfloat16().float())       # bf16-representable parameters: mode 1 is then exact
    ref = cpu_modules.ExtractorAttnCPU(C, k, torch.nn.LeakyReLU(0.1), softmax=True)
    ref.load_state_dict(m.state_dict())
    s, t = randn((B, C, H, W), seed=61).bfloat16(), randn((B, C, H, W), seed=62).bfloat16()
    f = make_flow("coherent", B, H, W, seed=63).bfloat16()
    up = randn((B, C, H, W), seed=64).bfloat16()
    m = m.to(DEV)
    a = [x.to(DEV).requires_grad_() for x in (s, t, f)]
    out = m(*a)
    assert out.dtype == torch.bfloat16
    out.backward(up.to(DEV))
    c = [x.float().requires_grad_() for x in (s, t, f)]
    want = ref(*c)
    want.backward(up.float())
    errs = {"out": rel_err(out.float().cpu(), want.detach())}
    for name, got, w in zip(("source", "target", "flow"), a, c):
        assert got.grad.dtype == torch.bfloat16
        errs[name] = rel_err(got.grad.float().cpu(), w.grad)
    for (n_, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        errs[n_] = rel_err(p.grad.float().cpu(), q.grad)
    # parameter gradients are sums over every position of softmax gradients built from the bf16-rounded attention
    # (heavy cancellation: 2^-9 relative noise per term against a sum much smaller than its terms): 2^-5 there
    bad = {n_: e for n_, e in errs.items() if e > (2 * tol if n_.startswith("fully_connect") else tol)}
    assert not bad, errs


@pytest.mark.parametrize("mode", (5, 4, 0))
@pytest.mark.parametrize("k,C", [(3, 16), (5, 8)])
def test_leaky_relu_at_exactly_zero_takes_the_negative_slope(lib, gfla, oracle, mode, k, C):
    """A hidden unit whose pre-activation is EXACTLY 0 everywhere (zero weights, zero bias): torch's LeakyReLU backward uses
    the negative slope at 0 (x > 0 ? 1 : slope), and so does csrc/fc_sample.hip / fc_tail.hip.  The bench-shape parity
    checks push their biases to +-8 to keep every unit off the kink; this is the one place where the value AT the kink is
    pinned: the gradients of that unit's bias and of everything behind it must be slope x (not 1 x, not 0 x) the upstream
    gradient, for both float32 FC arithmetic modes and the round-1 library path."""
    from oracle import cpu_modules
    B, H, W, slope, dead = 2, 12, 10, 0.1, 7
    torch.manual_seed(3)
    m = gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(slope), softmax=True)
    with torch.no_grad():
        # every other unit far from the kink (+-8, as the bench-shape tests), the pinned one exactly on it
        m.fully_connect_layer[0].bias.copy_(torch.where(torch.arange(128) % 2 == 0, 8.0, -8.0))
        m.fully_connect_layer[0].weight[dead].zero_()
        m.fully_connect_layer[0].bias[dead] = 0.0
    ref = cpu_modules.ExtractorAttnCPU(C, k, torch.nn.LeakyReLU(slope), softmax=True).double()
    ref.load_state_dict({n: v.double() for n, v in m.state_dict().items()})
    s, t = randn((B, C, H, W), seed=51), randn((B, C, H, W), seed=52)
    f = make_flow("coherent", B, H, W, seed=53)
    up = randn((B, C, H, W), seed=54)
    cargs = [x.double().clone().requires_grad_() for x in (s, t, f)]
    ref(*cargs).backward(up.double())
    want_b0 = ref.fully_connect_layer[0].bias.grad
    assert want_b0[dead].abs().item() > 1e-6          # the pinned unit does carry gradient: slope x something
    m = m.to(DEV)
    for impl in ("mfma", "library"):
        m.fc_impl, m.fc_mode = impl, mode
        args = [x.to(DEV).requires_grad_() for x in (s, t, f)]
        m.zero_grad()
        m(*args).backward(up.to(DEV))
        got_b0 = m.fully_connect_layer[0].bias.grad.cpu().double()
        assert abs(got_b0[dead].item() - want_b0[dead].item()) <= 1e-4 * want_b0.abs().max().item(), \
            "%s mode %d: d/d bias of the unit at the kink %.6e, reference %.6e (slope %.2f)" % (impl, mode, got_b0[dead], want_b0[dead], slope)
        for n_, g, w_ in zip(["source", "target", "flow"] + [n for n, _ in m.named_parameters()],
                             [a.grad.cpu() for a in args] + [p.grad.cpu() for p in m.parameters()],
                             [a.grad for a in cargs] + [p.grad for p in ref.parameters()]):
            assert_close(g, w_, 1e-4, "%s mode %d grad %s with a unit exactly on the kink" % (impl, mode, n_))


# ------------------------------------------------------------------------------- round 6: owner-computes scatter, f16 weight gradient
def _fc_backward_raw(lib, mode, s, t, f, w0, b0, w1, b1, up, k, slope=0.1):
    """gfla_fc_forward_f32 + gfla_fc_backward_f32 through the C ABI: every gradient the layer returns."""
    from global_flow_local_attention_amd import fc_mfma
    B, C, H, W = s.shape
    ws = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 0), dtype=torch.uint8, device=DEV)
    sc = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 1), dtype=torch.uint8, device=DEV)
    lg = torch.full((B, k * k, H, W), float("nan"), device=DEV)
    w1r = w1.reshape(k * k, 128).contiguous()
    lib.call("gfla_fc_forward_f32", s, _ptr(s), _ptr(t), _ptr(f), _ptr(w0), _ptr(b0), _ptr(w1r), _ptr(b1), _ptr(ws), _ptr(lg),
             B, C, H, W, k, slope, mode)
    gs, gt, gf = (torch.full_like(x, float("nan")) for x in (s, t, f))
    gw0, gb0, gw1, gb1 = (torch.full_like(x, float("nan")) for x in (w0, b0, w1r, b1))
    lib.call("gfla_fc_backward_f32", s, _ptr(ws), _ptr(f), _ptr(w1r), _ptr(up), _ptr(sc), _ptr(gs), _ptr(gt), _ptr(gf), _ptr(gw0),
             _ptr(gb0), _ptr(gw1), _ptr(gb1), B, C, H, W, k, slope, mode, 0)
    torch.cuda.synchronize()
    return lg, gs, gt, gf, gw0, gb0, gw1, gb1


def _collapse_flow(B, H, W):
    """Every position of a sample lands in ONE corner of the map (a gradient row that collects H W contributions: the list of
    fc_scatter_own_kernel needs several rounds, the fixed-point cells see their largest sums)."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    f = torch.stack((0.37 - xs, 0.21 - ys))[None].repeat(B, 1, 1, 1)
    f[1:] = f[1:] + 100.0   # the other samples: far outside, every tap clamps onto the last row / column
    return f.contiguous()


@pytest.mark.parametrize("mode", (5, 4, 0))
@pytest.mark.parametrize("k,C,H,W,kind", [(5, 16, 40, 28, "collapse"), (3, 16, 40, 28, "collapse"), (5, 8, 30, 22, "wild"),
                                          (3, 24, 33, 17, "smooth"), (5, 16, 64, 44, "coherent")])
def test_owner_computes_scatter_equals_the_atomics(lib, gfla, mode, k, C, H, W, kind):
    """fc_scatter_own_kernel (rows of the convolved source map's gradient owned by one workgroup, 64-bit fixed-point LDS cells)
    against round 2's global float atomics (tuning key 46 = 1) through the whole backward: the same gradients to float rounding
    of the SUMS (the atomics add in arrival order), incl. a flow that sends every position of a sample into one corner (> 1 024
    list entries for one workgroup: several list rounds) and far out-of-range flows; and bit-identical from run to run."""
    B = 3
    s, t = randn((B, C, H, W), seed=61).to(DEV), randn((B, C, H, W), seed=62).to(DEV)
    f = (_collapse_flow(B, H, W) if kind == "collapse" else make_flow(kind, B, H, W, seed=63)).to(DEV)
    w0 = (randn((128, 2 * C, k, k), seed=64) / (2 * C * k * k) ** 0.5).to(DEV)
    b0, b1 = (randn((128,), seed=65) * 0.1).to(DEV), (randn((k * k,), seed=67) * 0.1).to(DEV)
    w1 = (randn((k * k, 128, 1, 1), seed=66) / 128 ** 0.5).to(DEV)
    up = randn((B, k * k, H, W), seed=68).to(DEV)
    names = ("logits", "source", "target", "flow", "w0", "b0", "w1", "b1")
    own = _fc_backward_raw(lib, mode, s, t, f, w0, b0, w1, b1, up, k)
    again = _fc_backward_raw(lib, mode, s, t, f, w0, b0, w1, b1, up, k)
    old = gfla.set_tuning(46, 1)
    try:
        atom = _fc_backward_raw(lib, mode, s, t, f, w0, b0, w1, b1, up, k)
    finally:
        gfla.set_tuning(46, old)
    for n_, a, b_, c_ in zip(names, own, atom, again):
        assert torch.isfinite(a).all(), n_
        e = rel_err(a.cpu(), b_.cpu())
        # (the atomics add in arrival order: on the collapsing flow a cell collects > 1 000 float contributions and the
        # ATOMICS' own result moves by ~2e-6 from run to run; the owner-computes sums are exact integers, bit-equal below)
        assert e <= 1e-5, "%s: owner-computes vs atomics %.2e" % (n_, e)
        if n_ in ("source", "target", "flow", "b0", "w1", "b1", "logits"):   # (w0: the weight gradient's split sums are fixed order too)
            assert torch.equal(a, c_), "%s differs from run to run" % n_
    assert torch.equal(own[4], again[4]), "w0 differs from run to run"


def test_owner_computes_scatter_non_finite_gradient(lib, gfla):
    """An inf in the upstream gradient: the fixed-point cells cannot hold it -- the owned rows come out NaN (the reference's float
    atomics would poison only the cells the value reaches); nothing hangs, the other sample stays finite."""
    B, C, H, W, k = 2, 8, 12, 10, 5
    s, t = randn((B, C, H, W), seed=71).to(DEV), randn((B, C, H, W), seed=72).to(DEV)
    f = make_flow("smooth", B, H, W, seed=73).to(DEV)
    w0 = (randn((128, 2 * C, k, k), seed=74) / (2 * C * k * k) ** 0.5).to(DEV)
    b0, b1 = (randn((128,), seed=75) * 0.1).to(DEV), (randn((k * k,), seed=77) * 0.1).to(DEV)
    w1 = (randn((k * k, 128, 1, 1), seed=76) / 128 ** 0.5).to(DEV)
    up = randn((B, k * k, H, W), seed=78).to(DEV)
    up[0, 3, 5, 4] = float("inf")
    out = _fc_backward_raw(lib, 4, s, t, f, w0, b0, w1, b1, up, k)
    assert not torch.isfinite(out[1][0]).all()        # the poisoned sample's source gradient
    assert torch.isfinite(out[0]).all()               # the forward is untouched


@pytest.mark.parametrize("C,H,W", [(16, 64, 44), (24, 20, 26), (8, 9, 14), (16, 40, 66), (128, 64, 44)])
def test_two_term_f16_weight_gradient_equals_float32(lib, gfla, C, H, W):
    """fc_wino16_wgrad_kernel (mode 5, k = 5: both operands of the Winograd-domain products as two f16 terms on
    v_mfma_f32_16x16x32_f16) against the float32 kernel it replaces (tuning key 49 = 1) through the whole backward, one-row and
    multi-row units, and against float64 where the host can afford it (test_fc_function_against_float64 runs in mode 5 too)."""
    B, k = 2, 5
    s, t = randn((B, C, H, W), seed=81).to(DEV), randn((B, C, H, W), seed=82).to(DEV)
    f = make_flow("smooth", B, H, W, seed=83).to(DEV)
    w0 = (randn((128, 2 * C, k, k), seed=84) / (2 * C * k * k) ** 0.5).to(DEV)
    b0, b1 = (randn((128,), seed=85) * 0.1).to(DEV), (randn((k * k,), seed=87) * 0.1).to(DEV)
    w1 = (randn((k * k, 128, 1, 1), seed=86) / 128 ** 0.5).to(DEV)
    up = randn((B, k * k, H, W), seed=88).to(DEV)
    f16 = _fc_backward_raw(lib, 5, s, t, f, w0, b0, w1, b1, up, k)
    old = gfla.set_tuning(49, 1)
    try:
        f32 = _fc_backward_raw(lib, 5, s, t, f, w0, b0, w1, b1, up, k)
    finally:
        gfla.set_tuning(49, old)
    e = rel_err(f16[4].cpu(), f32[4].cpu())
    print("C %d %dx%d: f16 vs f32 weight gradient %.2e" % (C, H, W, e))
    assert torch.isfinite(f16[4]).all() and e <= 1e-5, e
    for a, b_ in zip(f16[:4] + f16[5:], f32[:4] + f32[5:]):
        assert torch.equal(a, b_)   # nothing else changes


@pytest.mark.parametrize("k,C,H,W", [(5, 16, 40, 28), (3, 24, 33, 17), (5, 128, 64, 44)])
def test_mode5_hybrid_dispatch_equals_the_winograd_form(lib, gfla, k, C, H, W):
    """Mode 5's default dispatch (direct f16x2 kernels reading the float32 maps in place for the k = 5 convolutions and every data
    gradient: fc_conv_kernel<..., SRC32>) against the all-Winograd form of the first half of round 6 (tuning key 52 = 1) through
    the whole layer: two float32-grade evaluations of the same sums, a few 1e-6 of the largest entry apart."""
    B = 2
    s, t = randn((B, C, H, W), seed=91).to(DEV), randn((B, C, H, W), seed=92).to(DEV)
    f = make_flow("smooth", B, H, W, seed=93).to(DEV)
    w0 = (randn((128, 2 * C, k, k), seed=94) / (2 * C * k * k) ** 0.5).to(DEV)
    b0, b1 = (randn((128,), seed=95) * 0.1).to(DEV), (randn((k * k,), seed=97) * 0.1).to(DEV)
    w1 = (randn((k * k, 128, 1, 1), seed=96) / 128 ** 0.5).to(DEV)
    up = randn((B, k * k, H, W), seed=98).to(DEV)
    hyb = _fc_backward_raw(lib, 5, s, t, f, w0, b0, w1, b1, up, k)
    old = gfla.set_tuning(52, 1)
    try:
        wino = _fc_backward_raw(lib, 5, s, t, f, w0, b0, w1, b1, up, k)
    finally:
        gfla.set_tuning(52, old)
    names = ("logits", "source", "target", "flow", "w0", "b0", "w1", "b1")
    for n_, a, b_ in zip(names, hyb, wino):
        e = rel_err(a.cpu(), b_.cpu())
        assert torch.isfinite(a).all() and e <= 1.5e-5, "%s: %.2e" % (n_, e)
