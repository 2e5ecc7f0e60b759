"""Parity of the DEFAULT dispatch at the BASELINE sizes on flows that are not smooth (round-3 review, item 1).

tests/test_bench_shapes_gpu.py pins the bench shapes on smooth flows only; here the same shapes meet

  wild          iid N(0, 8^2): patches land everywhere, many taps clamp at the border
  integer       exact integers: bilinear weights 0 / 1, every pixel a dense patch on a lattice point
  near_integer  integers +- 2^-22: the taps of a pixel stop forming a dense patch (tap-by-tap evaluation inside the
                dense-patch kernels, the non-dense lists of the matrix-core scatter)
  oob           every sample point far outside the map: all taps clamp onto one border row / column
                (block_extractor_kernel.cu:69-76 clamped index / unclamped weight; resample2d_kernel.cu:62-68)

for ExtractorAttn (FC arithmetic modes 4 = the Winograd default and 0), the standalone BlockExtractor (the lane-per-
flow-pixel forward kernel of round 4) and Resample2d (x176 AND the 256x256 VGG shapes), against the REAL reference
kernels (oracle/_ref) on the same GPU.  The reference runs in float64 except on near-integer flows, where a float32 and
a float64 evaluation legitimately floor a coordinate to different integers: there the reference kernels run in float32
(same expression order => same floor) for the ops, and the flow gradient of the attention block -- one-sided at a
lattice point -- is not compared.

Then the matrix-core scatter (csrc/patch_mfma.hip) of the two backward ops that can take it is forced ON, forced OFF and
driven across its device-side switch, each against the CPU oracle.
"""
import pytest
import torch
import torch.nn.functional as F

from test_bench_shapes_gpu import BENCH_SHAPES, NAMES, _ref, reference_extractor_attn, run_module
from util import assert_close, make_flow, randn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-5
KINDS = ("wild", "integer", "near_integer", "oob")


def rel_err(got, want, floor=0.0):
    """max |got - want| / max(max |want|, floor).  `floor` is for references that vanish identically: with every tap of
    every patch clamped onto one pixel the block's result no longer depends on the attention weights, so d/d logits and
    everything behind it (target, flow, FC parameters) is 0 in exact arithmetic and rounding noise (1e-17 in float64,
    1e-9 in float32) in practice -- those are held to an absolute bound, floor * tolerance."""
    want = want.detach().double()
    err = (got.detach().double().to(want.device) - want).abs().max().item()
    scale = max(want.abs().max().item(), floor)
    return err / scale if scale > 0 else err


def flow_of(kind, B, H, W, seed):
    if kind == "near_integer":
        f = make_flow("integer", B, H, W, seed=seed)
        return (f + torch.where(randn((B, 2, H, W), seed=seed + 1) > 0, 1.0, -1.0) * 2.0 ** -22).contiguous()
    if kind == "oob":
        f = make_flow("coherent", B, H, W, seed=seed)
        f[:, 0] += 1000.0   # x far to the right of the map
        f[:, 1] -= 1000.0   # y far above it
        # a quarter of the samples leave through the other two sides
        f[::4, 0] -= 2000.0
        f[::4, 1] += 2000.0
        return f.contiguous()
    return make_flow(kind, B, H, W, seed=seed)


# ------------------------------------------------------------------------------------------------ ExtractorAttn
def _case(B, C, H, W, k, kind, seed):
    s, t = randn((B, C, H, W), seed=seed), randn((B, C, H, W), seed=seed + 1)
    f = flow_of(kind, B, H, W, seed + 2)
    w0 = randn((128, 2 * C, k, k), seed=seed + 4) / (2 * C * k * k) ** 0.5
    # +-8 keeps every hidden activation off the LeakyReLU kink (tests/test_bench_shapes_gpu.py); out-of-bounds flows make
    # every source patch a constant, so the source half of the pre-activation is wider there: +-16
    mag = 16.0 if kind == "oob" else 8.0
    b0 = torch.where(torch.arange(128) % 2 == 0, mag, -mag) + randn((128,), seed=seed + 5) * 0.1
    w1 = randn((k * k, 128, 1, 1), seed=seed + 6) / 128 ** 0.5 * 0.3
    b1 = randn((k * k,), seed=seed + 7) * 0.1
    up = randn((B, C, H, W), seed=seed + 8)
    return s, t, f, w0, b0, w1, b1, up


def _reference(case, k, dtype):
    s, t, f, w0, b0, w1, b1, up = [x.to(dtype).to(DEV) for x in case]
    params = [p.clone().requires_grad_() for p in (w0, b0, w1, b1)]
    B = s.size(0)
    outs, gin, min_hidden = [], [[], [], []], float("inf")
    for lo in range(0, B, 8):
        a = [x[lo:lo + 8].clone().requires_grad_() for x in (s, t, f)]
        out, hidden = reference_extractor_attn(*a, *params, k)
        out.backward(up[lo:lo + 8])
        outs.append(out.detach())
        min_hidden = min(min_hidden, hidden.abs().min().item())
        for dst, x in zip(gin, a):
            dst.append(x.grad)
        del out, hidden
    torch.cuda.empty_cache()
    return torch.cat(outs), [torch.cat(g) for g in gin] + [p.grad for p in params], min_hidden


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("name,B,C,H,W,k", BENCH_SHAPES[:2])
def test_extractor_attn_bench_shape_rough_flows(gfla, name, B, C, H, W, k, kind):
    _ref()
    case = _case(B, C, H, W, k, kind, seed=900)
    want_out, want_grads, min_hidden = _reference(case, k, torch.float64)
    assert min_hidden > 1e-3, "test parameters put a hidden activation on the LeakyReLU kink (%.2e)" % min_hidden
    for mode in (5, 4, 0):
        out, grads = run_module(gfla, case, C, k, "mfma", mode)
        # oob: everything behind d/d logits vanishes identically (see rel_err); absolute bounds = tolerance x the scale the
        # tensor has on ordinary flows: 1e-2 per position, 100 for the parameter gradients (sums over 22 528 positions of
        # O(1e-1) softmax gradients x activations of magnitude 8-16)
        # (d/d flow is O(1..10) per position on ordinary flows -- bench.py's oracle check sees 6-8 -- so its floor is 1e-1:
        # the float32 sums over C channels x k*k taps that cancel to zero here leave ~2e-7 of rounding noise)
        floors = [({"source": 1e-2, "target": 1e-2, "flow": 1e-1}.get(n, 100.0)) if kind == "oob" else 0.0 for n in NAMES]
        errs = [("out", rel_err(out, want_out))] + [(n, rel_err(g, w, fl)) for n, g, w, fl in zip(NAMES, grads, want_grads, floors)]
        print("%s %s mode %d: " % (name, kind, mode) + " ".join("%s %.2e" % e for e in errs))
        for n, e in errs:
            if n == "flow" and kind in ("near_integer", "integer"):
                # d/d flow is one-sided on a lattice point; float32 and float64 may pick different sides (near_integer),
                # and on exact integers the 4-tap sampling of the convolved map (csrc/fc_sample.hip) and the reference's
                # per-tap derivative agree only as limits from the SAME side -- checked against the f32 chain below
                continue
            assert e <= TOL, "%s %s mode %d: %s rel err %.3e" % (name, kind, mode, n, e)
    if kind == "integer":
        # exact integers floor identically in float32: the float32 reference chain pins d/d flow too (looser: f32 chain)
        want_out, want_grads, _ = _reference(case, k, torch.float32)
        out, grads = run_module(gfla, case, C, k, "mfma", 4)
        e = rel_err(grads[2], want_grads[2])
        print("%s integer mode 4 d/d flow vs float32 reference chain: %.2e" % (name, e))
        assert e <= 1e-4, e


# ------------------------------------------------------------------------------------------------ BlockExtractor
@pytest.mark.parametrize("kind", ("smooth", "zero") + KINDS)
@pytest.mark.parametrize("name,B,C,H,W,k", BENCH_SHAPES)
def test_block_extractor_bench_shape_all_flows(gfla, name, B, C, H, W, k, kind):
    """Reference layout (B,C,kH,kW), forward + both gradients, default dispatch (forward: be_fwd_pix_kernel)."""
    if name.endswith("256x256") and kind not in ("smooth", "wild"):
        pytest.skip("the 256x256 shapes run the smooth and wild flows only (test time)")
    ref = _ref()
    from global_flow_local_attention_amd import _lib
    s = randn((B, C, H, W), seed=810)
    f = flow_of(kind, B, H, W, 811) if kind in KINDS else make_flow(kind, B, H, W, seed=811)
    sd, fd = s.to(DEV).requires_grad_(), f.to(DEV).requires_grad_()
    before = _lib.path_count(_lib.PATH_BE_FWD_PIX)
    out = gfla.BlockExtractor(k)(sd, fd)
    assert _lib.path_count(_lib.PATH_BE_FWD_PIX) == before + 1, "default dispatch did not take the lane-per-pixel kernel"
    up = randn((2, C, k * H, k * W), seed=812).to(DEV).repeat(B // 2, 1, 1, 1)
    out.backward(up)
    dt = torch.float32 if kind == "near_integer" else torch.float64
    sr, fr = s.to(dt).to(DEV), f.to(dt).to(DEV)
    want = ref.block_extractor_fwd(sr, fr, k)
    gs, gf = ref.block_extractor_bwd(sr, fr, up.to(dt), k)
    errs = (("out", rel_err(out, want)), ("grad source", rel_err(sd.grad, gs)), ("grad flow", rel_err(fd.grad, gf)))
    print("block_extractor %s %s: " % (name, kind) + " ".join("%s %.2e" % e for e in errs))
    for n, e in errs:
        # the float32 reference kernels sum thousands of float atomics in arbitrary order: 1e-4 there
        assert e <= (1e-4 if dt == torch.float32 and n != "out" else TOL), "block_extractor %s %s %s: rel err %.3e" % (name, kind, n, e)
    if kind == "zero":   # the reference's own KAT (test_block_extractor.py:46-55): centre tap = identity
        assert torch.equal(out[:, :, k // 2::k, k // 2::k], sd.detach())


def test_block_extractor_forward_kernels_agree(gfla):
    """The forward kernels -- round 4's wave-per-flow-row kernel (tuning key 0 = 4) and lane-per-pixel kernel with direct
    stores (3), round 1's lane-per-output-quad kernel (2) and the global-gather kernel (1) -- evaluate the reference's
    expression in the same order; the two round-4 kernels spell the multiply-adds out and must agree bit for bit, the
    older ones may differ by how the compiler contracted theirs, i.e. by an ulp of the largest term: 1e-6 absolute on
    unit-variance features."""
    from global_flow_local_attention_amd import _lib
    for (B, C, H, W, k) in ((2, 12, 64, 44, 5), (3, 10, 32, 22, 3), (2, 7, 19, 13, 4), (1, 5, 9, 6, 2), (2, 6, 24, 24, 5)):
        s = randn((B, C, H, W), seed=1).to(DEV)
        for kind in ("smooth", "wild", "near_integer"):
            f = (flow_of(kind, B, H, W, 2) if kind in KINDS else make_flow(kind, B, H, W, seed=2)).to(DEV)
            outs = []
            for key0 in (4, 3, 2, 1):
                old = gfla.set_tuning(0, key0)
                try:
                    o = torch.empty(B, C, k * H, k * W, device=DEV)
                    _lib.call("gfla_block_extractor_fwd_f32", s, _lib.ptr(s), _lib.ptr(f), _lib.ptr(o), B, C, H, W, H, W, k)
                    outs.append(o)
                finally:
                    gfla.set_tuning(0, old)
            assert torch.equal(outs[0], outs[1]), ((B, C, H, W, k), kind, "wave-per-row vs pixel kernel (same arithmetic: identical bits)")
            for o in outs[2:]:
                d = (outs[0] - o).abs().max().item()
                assert d <= 1e-6, ((B, C, H, W, k), kind, d)


GEOMETRIES = [("wrow", {0: 4, 4: 1, 24: 64}), ("wrow", {0: 4, 4: 3, 24: 192}), ("wrow", {0: 4, 4: 8, 24: 1024}), ("wrow", {0: 4, 10: 24}),
              ("pix", {0: 3, 4: 1, 5: 1, 24: 64}), ("pix", {0: 3, 4: 4, 5: 3, 24: 256}), ("pix", {0: 3, 4: 7, 5: 2, 24: 704}),
              ("pix", {0: 3, 4: 16, 5: 1, 24: 1024}), ("pix", {0: 3, 25: 1})]


@pytest.mark.parametrize("kernel,keys", GEOMETRIES)
def test_block_extractor_forward_kernel_geometries(gfla, oracle, kernel, keys):
    """Launch geometries of the two round-4 forward kernels that the default heuristics do not pick at the test shapes:
    wave-per-flow-row kernel (csrc/be_fwd_wrow.h: planes per workgroup, 1 .. 16 waves, several flow rows per wave with a
    ragged last group, a tight LDS budget), lane-per-pixel kernel (csrc/be_fwd_pix.h: planes incl. partial chunks, pixel splits with ragged last blocks, 1 .. 16
    waves, plain and non-temporal stores) -- against the CPU oracle, f32 and f64, Hs != Hf, odd widths (16-byte phase of
    the band's piece of the output plane: head / tail elements of the copy)."""
    from global_flow_local_attention_amd import _lib
    olds = {kk: gfla.set_tuning(kk, v) for kk, v in keys.items()}
    try:
        for (B, C, Hs, Ws, Hf, Wf, k) in ((2, 10, 21, 17, 21, 17, 3), (1, 6, 12, 20, 9, 14, 5), (2, 5, 8, 8, 11, 3, 4),
                                          (1, 3, 7, 5, 7, 5, 5), (1, 9, 16, 12, 16, 12, 2)):
            for dtype, tol_ in ((torch.float32, 2e-6), (torch.float64, 1e-12)):
                s = randn((B, C, Hs, Ws), dtype, seed=3)
                f = (make_flow("wild", B, Hf, Wf, seed=4) * 0.6).to(dtype)
                want = oracle.block_extractor_fwd(s, f, k)
                before = _lib.path_count(_lib.PATH_BE_FWD_PIX)
                # an output tensor that does not start on a 16-byte boundary: a view one element into a larger buffer
                buf = torch.full((want.numel() + 1,), float("nan"), dtype=dtype, device=DEV)
                got = buf[1:].view(want.shape)
                sd, fd = s.to(DEV), f.to(DEV)
                _lib.call("gfla_block_extractor_fwd_" + ("f32" if dtype == torch.float32 else "f64"), sd, _lib.ptr(sd),
                          _lib.ptr(fd), _lib.ptr(got), B, C, Hs, Ws, Hf, Wf, k)
                assert _lib.path_count(_lib.PATH_BE_FWD_PIX) == before + 1
                assert torch.isnan(buf[0]).item()
                assert_close(got.cpu(), want, tol_, "%s kernel %s keys %s" % (kernel, (B, C, Hs, Ws, Hf, Wf, k), keys))
    finally:
        for kk, v in olds.items():
            gfla.set_tuning(kk, v)


# ------------------------------------------------------------------------------------------------ Resample2d
@pytest.mark.parametrize("kind", ("smooth",) + KINDS)
@pytest.mark.parametrize("C,H,W", [(512, 32, 32), (256, 64, 64), (512, 32, 22), (256, 64, 44)])
def test_resample2d_bench_shape_all_flows(gfla, C, H, W, kind):
    """Resample2d(4, 1, sigma=2) forward + both gradients at the VGG-feature shapes of a 256x256 and a 256x176 image
    (bench.py legs / headline) against the real reference kernels (int() quirk of d/d input1 included: wild and
    out-of-bounds flows put x + dx < 0)."""
    if kind == "smooth" and W in (22, 44):
        pytest.skip("covered by tests/test_bench_shapes_gpu.py")
    ref = _ref()
    B = 32
    i1 = randn((B, C, H, W), seed=700)
    fl = flow_of(kind, B, H, W, 701) if kind in KINDS else make_flow(kind, B, H, W, seed=701)
    up = randn((B, C, H, W), seed=702)
    i1d, fld = i1.to(DEV).requires_grad_(), fl.to(DEV).requires_grad_()
    out = gfla.Resample2d(4, 1, 2)(i1d, fld)
    out.backward(up.to(DEV))
    # the tap set is truncated (4x4 support): a coordinate floored differently moves the whole support, so near-integer
    # flows are compared in the arithmetic type of the kernel under test
    dt = torch.float32 if kind == "near_integer" else torch.float64
    i2 = torch.cat((fl, torch.full((B, 1, H, W), 2.0)), 1).to(dt).to(DEV).contiguous()
    i1r = i1.to(dt).to(DEV)
    want = ref.resample2d_fwd(i1r, i2, 4, 1)
    g1, g2 = ref.resample2d_bwd(i1r, i2, up.to(dt).to(DEV), 4, 1)
    # oob: every tap clamps onto one pixel, the warp is that pixel whatever (dx, dy): d/d flow vanishes identically and
    # what is left is rounding noise of sums over C channels of O(1) terms -- absolute bound, tolerance x 1
    errs = (("out", rel_err(out, want)), ("grad input1", rel_err(i1d.grad, g1)),
            ("grad flow", rel_err(fld.grad, g2[:, :2], 1.0 if kind == "oob" else 0.0)))
    print("resample2d (%d,%d,%d,%d) %s: " % (B, C, H, W, kind) + " ".join("%s %.2e" % e for e in errs))
    for n, e in errs:
        assert e <= (1e-4 if dt == torch.float32 and n != "out" else TOL), "resample2d %s %s: rel err %.3e" % (kind, n, e)


def test_resample2d_tiny_sigma_fixed_point_planes(gfla, oracle):
    """sigma small enough for the far column weights to underflow: w_y / sum alone grows past 2^80 while every contribution
    w_y w_x / sum stays <= 1.  The fixed-point scatter forms the product in double (advisor finding, round 3).  Checked on
    the LDS-plane kernels (matrix-core scatter off, tuning key 14 = 1): fixed point (key 23 = 0) against round 1's double
    planes (key 23 = 1) and against the float32 oracle (the literal restatement of the reference's float kernel -- at these
    sigmas products of weights underflow in float32, so a float64 evaluation is a different function)."""
    from global_flow_local_attention_amd import _lib
    B, C, H, W = 2, 6, 24, 20
    i1 = randn((B, C, H, W), seed=1)
    go = randn((B, C, H, W), seed=2)
    o14 = gfla.set_tuning(14, 1)
    try:
        for sigma in (0.5, 0.09, 0.06, 0.045):
            flow = make_flow("smooth", B, H, W, seed=3) * 0.3
            i2 = torch.cat((flow, torch.full((B, 1, H, W), sigma)), 1).contiguous()
            want, _ = oracle.resample2d_bwd(i1, i2, go, 4, 1, True)
            i1d, i2d, god = i1.to(DEV), i2.to(DEV), go.to(DEV)     # (kept alive across the asynchronous calls)
            res = []
            for key23 in (0, 1):
                old = gfla.set_tuning(23, key23)
                try:
                    g1 = torch.full((B, C, H, W), float("nan"), device=DEV)
                    ws = _lib.scatter_workspace(g1, B, H, W, 16)
                    _lib.call("gfla_resample2d_bwd_ws_f32", g1, _lib.ptr(i1d), _lib.ptr(i2d), _lib.ptr(god),
                              _lib.ptr(g1), None, _lib.ptr(ws), B, C, H, W, H, W, 4, 1, 3)
                    torch.cuda.synchronize()
                    res.append(g1.cpu())
                finally:
                    gfla.set_tuning(23, old)
            scale = max(want[torch.isfinite(want)].abs().max().item(), 1e-3)
            bad = [int((~torch.isfinite(r)).sum()) for r in res] + [int((~torch.isfinite(want)).sum())]
            print("sigma %.3f: non-finite entries fixed / double / oracle %s; fixed vs double %.2e, fixed vs float32 oracle %.2e "
                  "(scale %.3g)" % (sigma, bad, (res[0] - res[1]).abs().nan_to_num(0).max().item(),
                                    (res[0] - want).abs().nan_to_num(0).max().item(), scale))
            # wherever the reference's float arithmetic itself stays finite, so must the fixed-point planes, with its values
            ok = torch.isfinite(want) & torch.isfinite(res[1])
            assert torch.isfinite(res[0][ok]).all(), sigma
            assert (res[0] - res[1])[ok].abs().max().item() <= 2e-5 * scale, sigma
            if sigma >= 0.06:   # below, the normalising sum sits in float32's denormal range: the host keeps denormals that
                                # the GPU's exp / multiply flush, and "the reference's float arithmetic" stops being one function
                assert (res[0] - want)[ok].abs().max().item() <= 1e-4 * scale, sigma
    finally:
        gfla.set_tuning(14, o14)


def test_resample2d_forward_small_sigma_stays_finite(gfla, oracle):
    """Forward with sigmas at which the Gaussian weight sum leaves float32's normal range (advisor finding, round 5: the
    reciprocal 1 / sum of the fast tap setup overflowed where the reference's val / sum is finite), and an integer flow with
    a sigma whose 2 sigma^2 is denormal / zero (0 * -inf in the exp2 form).  Wherever the float32 oracle is finite the kernel
    must be, and agree with it while the sum is a normal float (below, host denormals vs the GPU's flushes differ)."""
    B, C, H, W = 2, 5, 24, 20
    i1 = randn((B, C, H, W), seed=11)
    for sigma in (0.5, 0.09, 0.06, 0.05, 0.045, 1e-20, 0.0):
        for kind in ("smooth", "integer"):
            flow = make_flow(kind, B, H, W, seed=3) * (0.3 if kind == "smooth" else 1.0)
            i2 = torch.cat((flow, torch.full((B, 1, H, W), sigma)), 1).contiguous()
            want = oracle.resample2d_fwd(i1, i2, 4, 1)
            out = torch.full((B, C, H, W), float("nan"), device=DEV)
            i1d, i2d = i1.to(DEV), i2.to(DEV)
            from global_flow_local_attention_amd import _lib
            _lib.call("gfla_resample2d_fwd_f32", out, _lib.ptr(i1d), _lib.ptr(i2d), _lib.ptr(out), B, C, H, W, H, W, 4, 1)
            torch.cuda.synchronize()
            got = out.cpu()
            ok = torch.isfinite(want)
            assert torch.isfinite(got[ok]).all(), "sigma %g %s flow: %d non-finite outputs where the reference is finite" % (
                sigma, kind, int((~torch.isfinite(got[ok])).sum()))
            if sigma >= 0.06:
                assert (got - want)[ok].abs().max().item() <= 1e-4 * max(1.0, want[ok].abs().max().item()), (sigma, kind)


# ----------------------------------------------------------------- matrix-core scatter: on / off / across the switch
SCATTER_MODES = {           # tuning key 14 (0 auto, 1 never, 2 wherever supported), key 15 (rows per tile; >= 100000 = no limit)
    "auto": (0, 0),
    "forced_on": (2, 100000),
    "forced_off": (1, 0),
    "switch_bails_out": (2, 1),     # the device-side predicate always fails: the matrix-core kernels return at once and the
                                    # LDS-atomic kernel launched behind them (inverse predicate) does the work
}


def _with_scatter_mode(gfla, mode, fn):
    k14, k15 = SCATTER_MODES[mode]
    o14, o15 = gfla.set_tuning(14, k14), gfla.set_tuning(15, k15)
    try:
        return fn()
    finally:
        gfla.set_tuning(14, o14)
        gfla.set_tuning(15, o15)


@pytest.mark.parametrize("mode", list(SCATTER_MODES))
@pytest.mark.parametrize("shape,k", [((2, 24, 64, 44), 5), ((3, 40, 32, 22), 3), ((2, 130, 32, 32), 3)])
def test_aggregate_source_gradient_scatter_paths(gfla, oracle, mode, shape, k):
    """d/d source (+ d/d flow, d/d logits) of the aggregation through gfla_local_attn_aggregate_bwd_ws_f32 with the
    matrix-core scatter forced on, forced off, bailing out on the device, and in the automatic dispatch over flow scales
    that straddle its 2.6x rows-per-tile switch (smooth x 0.3 ... wild): each against autograd through the float64
    gather formulation (exact for these gradients)."""
    from global_flow_local_attention_amd import _lib
    B, C, H, W = shape
    s = randn((B, C, H, W), seed=21)
    lg = randn((B, k * k, H, W), seed=22) * 2
    go = randn((B, C, H, W), seed=23)
    for tag, f in (("smooth x0.3", make_flow("smooth", B, H, W, seed=24) * 0.3), ("smooth", make_flow("smooth", B, H, W, seed=24)),
                   ("smooth x2.5", make_flow("smooth", B, H, W, seed=24) * 2.5), ("smooth x6", make_flow("smooth", B, H, W, seed=24) * 6),
                   ("wild", make_flow("wild", B, H, W, seed=24)), ("near_integer", flow_of("near_integer", B, H, W, 24))):
        s64, f64, l64 = s.double().requires_grad_(), f.double().requires_grad_(), lg.double().requires_grad_()
        a = torch.softmax(l64, 1)
        ref = F.avg_pool2d(F.pixel_shuffle(a, k) * oracle.block_extractor_gather(s64, f64, k), k, k)
        ref.backward(go.double())
        sd, fd, ad, god = s.to(DEV), f.to(DEV), a.detach().float().to(DEV).contiguous(), go.to(DEV)

        def run():
            gs, gf, gl = torch.zeros_like(sd), torch.zeros_like(fd), torch.zeros_like(ad)
            ws = _lib.scatter_workspace(sd, B, H, W, (k + 1) ** 2)
            _lib.call("gfla_local_attn_aggregate_bwd_ws_f32", sd, _lib.ptr(sd), _lib.ptr(fd), _lib.ptr(ad), _lib.ptr(god),
                      _lib.ptr(gs), _lib.ptr(gf), _lib.ptr(gl), _lib.ptr(ws), B, C, H, W, H, W, k, 1)
            torch.cuda.synchronize()
            return gs, gf, gl
        gs, gf, gl = _with_scatter_mode(gfla, mode, run)
        what = "%s %s k%d %s" % (mode, shape, k, tag)
        assert_close(gs.cpu(), s64.grad.float(), 2e-5, "grad_source " + what)
        assert_close(gl.cpu(), l64.grad.float(), 2e-5, "grad_logits " + what)
        if tag != "near_integer":
            assert_close(gf.cpu(), f64.grad.float(), 4e-5, "grad_flow " + what)


@pytest.mark.parametrize("mode", list(SCATTER_MODES))
@pytest.mark.parametrize("shape", [(2, 40, 32, 22), (2, 24, 64, 44), (1, 130, 32, 32)])
def test_resample2d_input1_gradient_scatter_paths(gfla, oracle, mode, shape):
    """d/d input1 of Resample2d(4, 1) through gfla_resample2d_bwd_ws_f32 (reference int() quirk on), same four dispatch
    modes and flow scales, against the literal float64 restatement of resample2d_kernel.cu:98-202."""
    from global_flow_local_attention_amd import _lib
    B, C, H, W = shape
    i1 = randn((B, C, H, W), seed=31)
    go = randn((B, C, H, W), seed=32)
    for tag, f in (("smooth x0.3", make_flow("smooth", B, H, W, seed=33) * 0.3), ("smooth", make_flow("smooth", B, H, W, seed=33)),
                   ("smooth x2.5", make_flow("smooth", B, H, W, seed=33) * 2.5), ("smooth x6", make_flow("smooth", B, H, W, seed=33) * 6),
                   ("wild", make_flow("wild", B, H, W, seed=33)), ("oob", flow_of("oob", B, H, W, 33))):
        i2 = torch.cat((f, torch.full((B, 1, H, W), 2.0)), 1).contiguous()
        want, _ = oracle.resample2d_bwd(i1.double(), i2.double(), go.double(), 4, 1, True)
        i1d, i2d, god = i1.to(DEV), i2.to(DEV), go.to(DEV)

        def run():
            g1 = torch.full_like(i1d, float("nan"))
            ws = _lib.scatter_workspace(i1d, B, H, W, 16)
            _lib.call("gfla_resample2d_bwd_ws_f32", i1d, _lib.ptr(i1d), _lib.ptr(i2d), _lib.ptr(god), _lib.ptr(g1), None,
                      _lib.ptr(ws), B, C, H, W, H, W, 4, 1, 3)
            torch.cuda.synchronize()
            return g1
        g1 = _with_scatter_mode(gfla, mode, run)
        assert_close(g1.cpu(), want.float(), 2e-5, "grad_input1 %s %s %s" % (mode, shape, tag))
