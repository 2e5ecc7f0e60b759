"""Winograd-domain FC kernels (arithmetic modes 4, csrc/fc_wino.hip, and 5, csrc/fc_wino16.hip: the same domain with two-term
f16 operands on the f16 matrix cores, held to the same bars) beyond the shapes test_fc_mfma_gpu.py /
test_bench_shapes_gpu.py already run in every mode: a sweep over ragged geometries (partial tiles in both directions,
maps smaller than one tile group, channels that do not fill a chunk, single samples, maps that need the single-buffer
staging) against float64 convolutions on the host, the mode-4 -> mode-0 fallback, and size-independent properties at the
bench shape (linearity of each convolution in its input, adjointness of forward and data gradient).

Bars: the same as every other arithmetic mode (forward 1e-5, gradients 2e-5 of the largest reference entry)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from util import max_abs, randn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FWD_TOL, GRAD_TOL = 1e-5, 2e-5


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def rel_err(got, want):
    return max_abs(got, want) / max(1e-30, want.double().abs().max().item())


def _pads(k, is_source):
    lo, hi = k // 2, k - 1 - k // 2
    return (k - 1, k - 1, k - 1, k - 1) if is_source else (lo, hi, lo, hi)


def _run_half(B, C, H, W, k, is_source, mode, x, w0, dG):
    from global_flow_local_attention_amd import _lib, fc_mfma
    g = fc_mfma.geometry(H, W, k, is_source)
    ws = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 0), dtype=torch.uint8, device=DEV)
    out = torch.full((B, g["Mg"], 128), float("nan"), device=DEV)
    _lib.call("gfla_fc_conv_fwd_f32", x, _ptr(x), _ptr(w0), is_source, _ptr(ws), _ptr(out), B, C, H, W, k, mode)
    rows = (torch.arange(g["Ho"])[:, None] * g["Wp"] + torch.arange(g["Wo"])[None, :]).reshape(-1).to(DEV)
    z = torch.zeros(B, g["Sz"], 128, device=DEV)
    z[:, g["lead"] + rows, :] = dG.permute(0, 2, 3, 1).reshape(B, -1, 128)
    sc = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 1), dtype=torch.uint8, device=DEV)
    gx = torch.full((B, C, H, W), float("nan"), device=DEV)
    gw = torch.full((128, 2 * C, k, k), float("nan"), device=DEV)
    _lib.call("gfla_fc_conv_bwd_f32", x, _ptr(z), is_source, _ptr(ws), _ptr(sc), _ptr(gx), _ptr(gw), B, C, H, W, k, mode)
    fwd = out[:, :g["Ho"] * g["Wo"], :].reshape(B, g["Ho"], g["Wo"], 128).permute(0, 3, 1, 2)
    return fwd, gx, gw, g


SWEEP = [  # (k, B, C, H, W): ragged / tiny / odd / wide
    (5, 1, 5, 3, 3), (5, 3, 17, 7, 5), (5, 2, 16, 2, 9), (5, 1, 33, 13, 31), (5, 2, 8, 40, 66), (5, 1, 16, 9, 120),
    (3, 1, 5, 3, 3), (3, 3, 17, 7, 5), (3, 2, 16, 2, 9), (3, 1, 40, 13, 31), (3, 2, 8, 33, 65), (3, 1, 16, 10, 200),
]


@pytest.mark.parametrize("k,B,C,H,W", SWEEP)
@pytest.mark.parametrize("is_source", [0, 1])
@pytest.mark.parametrize("wmode", [4, 5, 52, 50])
def test_winograd_half_on_ragged_shapes(gfla, k, B, C, H, W, is_source, wmode):
    """wmode 5 = the default dispatch of mode 5 (later in round 6: the direct f16x2 kernels fed from the float32 maps for the
    k = 5 convolutions and every data gradient, Winograd domain for the k = 3 forward and the weight gradients); 52 = mode 5
    with Winograd-domain kernels for every convolution (tuning key 52 = 1: the first half of round 6); 50 = that with the
    two-term f16 kernel forced for the k = 3 data gradient as well (key 43 = 1; the float32 Winograd kernel otherwise)."""
    from global_flow_local_attention_amd import fc_mfma
    force16, all_wino = wmode == 50, wmode in (50, 52)
    wmode = 5 if all_wino else wmode
    if force16 and k != 3:
        pytest.skip("key 43 only changes k = 3")
    old43, old52 = gfla.set_tuning(43, 1 if force16 else 0), gfla.set_tuning(52, 1 if all_wino else 0)
    try:
        _ragged(gfla, k, B, C, H, W, is_source, wmode)
    finally:
        gfla.set_tuning(43, old43)
        gfla.set_tuning(52, old52)


def _ragged(gfla, k, B, C, H, W, is_source, wmode):
    from global_flow_local_attention_amd import fc_mfma
    mode = fc_mfma.resolve_mode(C, H, W, k, wmode)
    assert mode in (wmode, 4, 0)
    if wmode == 5 and mode != 5:
        pytest.skip("shape falls back (covered by the mode-4 row)")
    x = (randn((B, C, H, W), seed=1) * 1.7).to(DEV)
    w0 = (randn((128, 2 * C, k, k), seed=2) * 0.05).to(DEV)
    g = fc_mfma.geometry(H, W, k, is_source)
    dG = (randn((B, 128, g["Ho"], g["Wo"]), seed=3) * 1e-3).to(DEV)
    fwd, gx, gw, _ = _run_half(B, C, H, W, k, is_source, mode, x, w0, dG)
    x64 = x.cpu().double().requires_grad_()
    wh = (w0[:, C:] if is_source else w0[:, :C]).cpu().double().clone().requires_grad_()
    ref = F.conv2d(F.pad(x64, _pads(k, is_source), mode="replicate"), wh)
    ref.backward(dG.cpu().double())
    e_f, e_x = rel_err(fwd.cpu(), ref.detach()), rel_err(gx.cpu(), x64.grad)
    e_w = rel_err((gw[:, C:] if is_source else gw[:, :C]).cpu(), wh.grad)
    print("mode %d k %d B %d C %d %dx%d half %d: map %.2e grad_x %.2e grad_w %.2e" % (mode, k, B, C, H, W, is_source, e_f, e_x, e_w))
    assert e_f <= FWD_TOL and e_x <= GRAD_TOL and e_w <= GRAD_TOL, (e_f, e_x, e_w)
    other = gw[:, :C] if is_source else gw[:, C:]
    assert float(other.abs().max()) == 0.0


def test_mode4_falls_back_to_the_direct_kernels_where_its_tiles_do_not_fit(gfla):
    """A map too wide for the Winograd kernel's LDS span: resolve_mode answers 0, the module runs the direct f32 kernels
    and still matches the reference's op-by-op composition on the library's own ops."""
    from global_flow_local_attention_amd import _lib, fc_mfma
    C, H, W, k = 8, 6, 200, 5
    assert not fc_mfma.supported(C, H, W, k, 4) and not fc_mfma.supported(C, H, W, k, 5) and fc_mfma.resolve_mode(C, H, W, k) == 0
    m = gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True).to(DEV)
    s, t = randn((1, C, H, W), seed=1).to(DEV), randn((1, C, H, W), seed=2).to(DEV)
    f = (randn((1, 2, H, W), seed=3) * 1.5).to(DEV)
    before = _lib.path_count(_lib.PATH_FC_FWD_MODE0)
    a = m(s, t, f)
    assert _lib.path_count(_lib.PATH_FC_FWD_MODE0) == before + 1
    m.fused = False
    assert rel_err(a.cpu(), m(s, t, f).cpu()) <= 2e-5


@pytest.mark.parametrize("scale_x,scale_w,scale_g", [(1e-18, 1.0, 1.0), (3e12, 1e-9, 1e-25), (1.0, 7e7, 1e9), (2e-30, 3e-8, 1e-6)])
@pytest.mark.parametrize("k,B,C,H,W", [(5, 2, 24, 11, 9), (3, 2, 40, 9, 14)])
def test_two_term_f16_operands_over_the_float32_range(gfla, k, B, C, H, W, scale_x, scale_w, scale_g):
    """Mode 5 splits every transformed value into two f16 terms after scaling by a power of two taken from the tensor's max |x|:
    inputs, weights and gradients of any float32 magnitude must come out as accurately as O(1) ones (the scales are exact, the
    inverse scales are applied as two power-of-two factors).  Magnitudes whose PRODUCTS leave the float32 range are not
    float32 problems and are not tested."""
    from global_flow_local_attention_amd import fc_mfma
    assert fc_mfma.resolve_mode(C, H, W, k, 5) == 5
    for is_source in (0, 1):
        x = (randn((B, C, H, W), seed=1) * scale_x).to(DEV)
        w0 = (randn((128, 2 * C, k, k), seed=2) * scale_w).to(DEV)
        g = fc_mfma.geometry(H, W, k, is_source)
        dG = (randn((B, 128, g["Ho"], g["Wo"]), seed=3) * scale_g).to(DEV)
        fwd, gx, gw, _ = _run_half(B, C, H, W, k, is_source, 5, x, w0, dG)
        x64 = x.cpu().double().requires_grad_()
        wh = (w0[:, C:] if is_source else w0[:, :C]).cpu().double().clone().requires_grad_()
        ref = F.conv2d(F.pad(x64, _pads(k, is_source), mode="replicate"), wh)
        ref.backward(dG.cpu().double())
        e_f, e_x = rel_err(fwd.cpu(), ref.detach()), rel_err(gx.cpu(), x64.grad)
        e_w = rel_err((gw[:, C:] if is_source else gw[:, :C]).cpu(), wh.grad)
        print("scales %g %g %g k %d half %d: map %.2e grad_x %.2e grad_w %.2e" % (scale_x, scale_w, scale_g, k, is_source, e_f, e_x, e_w))
        assert e_f <= FWD_TOL and e_x <= GRAD_TOL and e_w <= GRAD_TOL, (e_f, e_x, e_w)


@pytest.mark.parametrize("k,B,C,H,W", [(5, 2, 24, 11, 9), (3, 2, 40, 9, 14)])
def test_two_term_f16_operands_all_zero_and_denormal_tensors(gfla, k, B, C, H, W):
    """max |x| = 0 (no scale can be derived: scale 1) and max |x| in float32's denormal range: mode 5 must return exact zeros /
    finite values, never NaN from a 0 * inf or an overflowing inverse scale."""
    from global_flow_local_attention_amd import fc_mfma
    for scale_x, scale_w, scale_g in ((0.0, 1.0, 1.0), (1.0, 0.0, 1.0), (1.0, 1.0, 0.0), (1e-42, 1.0, 1e-43)):
        for is_source in (0, 1):
            x = (randn((B, C, H, W), seed=1) * scale_x).to(DEV)
            w0 = (randn((128, 2 * C, k, k), seed=2) * 0.05 * scale_w).to(DEV)
            g = fc_mfma.geometry(H, W, k, is_source)
            dG = (randn((B, 128, g["Ho"], g["Wo"]), seed=3) * scale_g).to(DEV)
            fwd, gx, gw, _ = _run_half(B, C, H, W, k, is_source, 5, x, w0, dG)
            for name, t in (("map", fwd), ("grad_x", gx), ("grad_w", gw)):
                assert torch.isfinite(t).all(), (scale_x, scale_w, scale_g, name)
            if scale_x == 0.0 or scale_w == 0.0:
                assert float(fwd.abs().max()) == 0.0
            if scale_g == 0.0 or scale_w == 0.0:
                assert float(gx.abs().max()) == 0.0
            if scale_g == 0.0 or scale_x == 0.0:
                assert float(gw.abs().max()) == 0.0


@pytest.mark.parametrize("wmode", [4, 5])
@pytest.mark.parametrize("k,C,H,W", [(5, 128, 64, 44), (3, 256, 32, 22)])
def test_winograd_linearity_and_adjointness_at_bench_shape(gfla, k, C, H, W, wmode):
    """Properties that need no reference at B = 32: conv(a x1 + b x2) = a conv(x1) + b conv(x2);  <conv(x), dG> = <x, convT(dG)>
    (forward and data gradient are each other's adjoints);  <dW, W> = <conv_W(x), dG> (weight gradient)."""
    from global_flow_local_attention_amd import fc_mfma
    B, is_source = 32, 1
    g = fc_mfma.geometry(H, W, k, is_source)
    x1, x2 = randn((B, C, H, W), seed=11).to(DEV), randn((B, C, H, W), seed=12).to(DEV)
    w0 = (randn((128, 2 * C, k, k), seed=13) / (2 * C * k * k) ** 0.5).to(DEV)
    dG = randn((B, 128, g["Ho"], g["Wo"]), seed=14).to(DEV)
    f1, gx, gw, _ = _run_half(B, C, H, W, k, is_source, wmode, x1, w0, dG)
    f2, _, _, _ = _run_half(B, C, H, W, k, is_source, wmode, x2, w0, dG)
    f12, _, _, _ = _run_half(B, C, H, W, k, is_source, wmode, 0.75 * x1 - 1.5 * x2, w0, dG)
    assert rel_err(f12, 0.75 * f1 - 1.5 * f2) <= 2e-5
    lhs = (f1.double() * dG.double()).sum().item()
    rhs = (x1.double() * gx.double()).sum().item()
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), (f1.double().abs() * dG.double().abs()).sum().item() * 1e-3)
    wlhs = (gw[:, C:].double() * w0[:, C:].double()).sum().item()
    assert abs(wlhs - lhs) <= 2e-5 * max(abs(lhs), (f1.double().abs() * dG.double().abs()).sum().item() * 1e-3)


@pytest.mark.parametrize("mode", [4, 5])
@pytest.mark.parametrize("k,B,C,H,W", [(5, 3, 24, 11, 9), (3, 2, 40, 9, 14), (5, 2, 128, 64, 44)])
def test_two_job_launch_equals_one_launch_per_half(gfla, k, B, C, H, W, mode):
    """gfla_fc_forward / gfla_fc_backward issue the convolutions of the target and the source half as ONE launch (workgroups
    [0, n0) = job 0, the rest job 1; csrc/fc_wino.hip).  Tuning key 21 = 2 launches them separately: every output must be
    bit-identical (the jobs share nothing but the grid)."""
    from global_flow_local_attention_amd import _lib, fc_mfma
    if fc_mfma.resolve_mode(C, H, W, k, mode) != mode:
        pytest.skip("shape falls back to the direct kernels")
    s, t = randn((B, C, H, W), seed=1).to(DEV), randn((B, C, H, W), seed=2).to(DEV)
    f = (randn((B, 2, H, W), seed=3) * 1.5).to(DEV)
    w0, w1 = (randn((128, 2 * C, k, k), seed=4) * 0.05).to(DEV), (randn((k * k, 128), seed=5) * 0.1).to(DEV)
    gl = (randn((B, k * k, H, W), seed=6) * 1e-2).to(DEV)

    def run():
        ws = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 0), dtype=torch.uint8, device=DEV)
        sc = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 1), dtype=torch.uint8, device=DEV)
        logits = torch.empty(B, k * k, H, W, device=DEV)
        gs, gt, gf, gw0 = torch.zeros_like(s), torch.empty_like(t), torch.zeros_like(f), torch.empty_like(w0)
        _lib.call("gfla_fc_forward_f32", s, _ptr(s), _ptr(t), _ptr(f), _ptr(w0), None, _ptr(w1), None, _ptr(ws), _ptr(logits),
                  B, C, H, W, k, 0.1, mode)
        _lib.call("gfla_fc_backward_f32", s, _ptr(ws), _ptr(f), _ptr(w1), _ptr(gl), _ptr(sc), _ptr(gs), _ptr(gt), _ptr(gf),
                  _ptr(gw0), None, None, None, B, C, H, W, k, 0.1, mode, 0)
        torch.cuda.synchronize()
        return logits, gs, gt, gw0

    merged = run()
    old = gfla.set_tuning(21, 2)
    try:
        separate = run()
    finally:
        gfla.set_tuning(21, old)
    for name, a, b in zip(("logits", "grad_source", "grad_target", "grad_w0"), merged, separate):
        if name in ("grad_source", "grad_w0"):  # behind the splat into the source map's gradient: float atomics, whose
            assert rel_err(a, b) < 1e-5, name   # order moves the last bits from run to run (1.2e-6 seen)
        else:
            assert torch.equal(a, b), name


WGRAD_SWEEP = SWEEP + [(3, 4, 32, 32, 22), (3, 2, 24, 12, 30), (5, 2, 16, 20, 26), (5, 3, 8, 9, 14), (3, 2, 16, 31, 7)]


@pytest.mark.parametrize("k,B,C,H,W", WGRAD_SWEEP)
@pytest.mark.parametrize("is_source", [0, 1])
@pytest.mark.parametrize("form", ["winograd", "winograd_units_of_16_tiles", "winograd_single_row_units"])
def test_winograd_domain_weight_gradient_forms(gfla, k, B, C, H, W, is_source, form):
    """The Winograd-domain weight gradient forced for every k (tuning key 19 = 2; the default takes it for k = 5 and wherever
    multi-row units apply) with units of whole tile rows on narrow maps (the default: up to 32 tiles; key 29 = 2: up to 16)
    and with round 3's single-row units (key 29 = 1), against float64 on the host."""
    from global_flow_local_attention_amd import fc_mfma
    if fc_mfma.resolve_mode(C, H, W, k, 4) != 4:
        pytest.skip("shape falls back to the direct kernels")
    x = (randn((B, C, H, W), seed=5) * 1.3).to(DEV)
    w0 = (randn((128, 2 * C, k, k), seed=6) * 0.05).to(DEV)
    g = fc_mfma.geometry(H, W, k, is_source)
    dG = (randn((B, 128, g["Ho"], g["Wo"]), seed=7) * 1e-3).to(DEV)
    old19, old29 = gfla.set_tuning(19, 2), gfla.set_tuning(29, {"winograd": 0, "winograd_units_of_16_tiles": 2, "winograd_single_row_units": 1}[form])
    try:
        _, _, gw, _ = _run_half(B, C, H, W, k, is_source, 4, x, w0, dG)
        torch.cuda.synchronize()
    finally:
        gfla.set_tuning(19, old19)
        gfla.set_tuning(29, old29)
    x64 = x.cpu().double()
    wh = (w0[:, C:] if is_source else w0[:, :C]).cpu().double().clone().requires_grad_()
    F.conv2d(F.pad(x64, _pads(k, is_source), mode="replicate"), wh).backward(dG.cpu().double())
    e_w = rel_err((gw[:, C:] if is_source else gw[:, :C]).cpu(), wh.grad)
    print("k %d B %d C %d %dx%d half %d %s: grad_w %.2e" % (k, B, C, H, W, is_source, form, e_w))
    assert e_w <= GRAD_TOL, e_w
