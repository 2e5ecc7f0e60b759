"""TEST INFRASTRUCTURE ONLY -- Python face of the CPU oracle (oracle/gfla_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.

Two layers:

* literal layer  -- ctypes calls into libgfla_oracle.so, the per-index C
  restatement of the seven reference kernels (fp32 and fp64, reference quirks
  included, `trunc_compat` switch for resample2d_kernel.cu:137-138);
* identity layer -- independent torch formulations (pixel_shuffle, grid_sample,
  gather) that do NOT come from the reference and are used to cross-check the
  literal layer and to obtain autograd gradients.

All functions take and return contiguous CPU torch tensors.
"""
import ctypes
import os
import subprocess

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgfla_oracle.so")
_lib = None


def build(force=False):
    """Compile libgfla_oracle.so with gcc (seconds)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "libgfla_oracle.so"] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def set_threads(n):
    """Number of host threads the oracle's OpenMP loops use."""
    omp = ctypes.CDLL("libgomp.so.1")
    omp.omp_set_num_threads(int(n))


def _sfx(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError("oracle supports float32/float64, got %s" % t.dtype)


def _p(t):
    assert t.device.type == "cpu" and t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def _i64(v):
    return ctypes.c_int64(int(v))


# --------------------------------------------------------------------------- literal layer
def block_extractor_fwd(source, flow, k):
    """block_extractor_kernel.cu:20-85 via block_extractor.py:8-28."""
    B, C, Hs, Ws = source.shape
    Bf, two, Hf, Wf = flow.shape
    assert two == 2 and Bf == B
    out = torch.zeros(B, C, k * Hf, k * Wf, dtype=source.dtype)
    fn = getattr(lib(), "oracle_block_extractor_fwd_" + _sfx(source))
    fn(_p(source), _p(flow), _p(out), _i64(B), _i64(C), _i64(Hs), _i64(Ws), _i64(Hf), _i64(Wf),
       ctypes.c_int(k))
    return out


def block_extractor_bwd(source, flow, grad_out, k):
    """block_extractor_kernel.cu:89-170 via block_extractor.py:31-42."""
    B, C, Hs, Ws = source.shape
    _, _, Hf, Wf = flow.shape
    gs = torch.zeros_like(source)
    gf = torch.zeros_like(flow)
    fn = getattr(lib(), "oracle_block_extractor_bwd_" + _sfx(source))
    fn(_p(source), _p(flow), _p(grad_out.contiguous()), _p(gs), _p(gf),
       _i64(B), _i64(C), _i64(Hs), _i64(Ws), _i64(Hf), _i64(Wf), ctypes.c_int(k))
    return gs, gf


def local_attn_reshape_fwd(inputs, k):
    """local_attn_reshape_kernel.cu:20-61 via local_attn_reshape.py:8-25."""
    B, C, H, W = inputs.shape
    assert C == k * k
    out = torch.zeros(B, 1, k * H, k * W, dtype=inputs.dtype)
    fn = getattr(lib(), "oracle_local_attn_reshape_fwd_" + _sfx(inputs))
    fn(_p(inputs), _p(out), _i64(B), _i64(H), _i64(W), ctypes.c_int(k))
    return out


def local_attn_reshape_bwd(grad_out, k):
    """local_attn_reshape_kernel.cu:65-108 via local_attn_reshape.py:28-37."""
    B, one, Ho, Wo = grad_out.shape
    H, W = Ho // k, Wo // k
    gi = torch.zeros(B, k * k, H, W, dtype=grad_out.dtype)
    fn = getattr(lib(), "oracle_local_attn_reshape_bwd_" + _sfx(grad_out))
    fn(_p(grad_out.contiguous()), _p(gi), _i64(B), _i64(H), _i64(W), ctypes.c_int(k))
    return gi


def resample2d_fwd(input1, input2, k=2, dilation=1):
    """resample2d_kernel.cu:20-95 via resample2d.py:9-23; input2 is (B,3,H,W)."""
    _, C, Hi, Wi = input1.shape
    B, three, H, W = input2.shape
    assert three == 3
    out = torch.zeros(B, C, H, W, dtype=input1.dtype)
    fn = getattr(lib(), "oracle_resample2d_fwd_" + _sfx(input1))
    fn(_p(input1), _p(input2), _p(out), _i64(B), _i64(C), _i64(Hi), _i64(Wi), _i64(H), _i64(W),
       ctypes.c_int(k), ctypes.c_int(dilation))
    return out


def resample2d_bwd(input1, input2, grad_out, k=2, dilation=1, trunc_compat=True):
    """resample2d_kernel.cu:98-202 and :204-330 via resample2d.py:26-39.

    trunc_compat=True reproduces the reference's int() truncation in the
    input1 gradient; False gives the gradient that matches the forward pass.
    """
    _, C, Hi, Wi = input1.shape
    B, _, H, W = input2.shape
    g1 = torch.zeros_like(input1)
    g2 = torch.zeros_like(input2)
    go = grad_out.contiguous()
    sfx = _sfx(input1)
    f1 = getattr(lib(), "oracle_resample2d_bwd_input1_" + sfx)
    f1(_p(input2), _p(go), _p(g1), _i64(B), _i64(C), _i64(Hi), _i64(Wi), _i64(H), _i64(W),
       ctypes.c_int(k), ctypes.c_int(dilation), ctypes.c_int(1 if trunc_compat else 0))
    f2 = getattr(lib(), "oracle_resample2d_bwd_input2_" + sfx)
    f2(_p(input1), _p(input2), _p(go), _p(g2), _i64(B), _i64(C), _i64(Hi), _i64(Wi), _i64(H), _i64(W),
       ctypes.c_int(k), ctypes.c_int(dilation))
    return g1, g2


def resample2d_module_fwd(input1, flow, k=2, dilation=1, sigma=5.0):
    """Resample2d.forward (resample2d.py:49-53): appends the constant sigma channel."""
    B, _, H, W = flow.shape
    sig = torch.full((B, 1, H, W), float(sigma), dtype=flow.dtype)
    return resample2d_fwd(input1.contiguous(), torch.cat((flow, sig), 1).contiguous(), k, dilation)


def extractor_attn_fwd(source, target, flow, w0, b0, w1, b1, k, negative_slope=0.1,
                       return_attn=False):
    """ExtractorAttn.forward / hook_attn_param (base_function.py:804-818) with the
    literal-layer ops; the two convolutions and the softmax are stock torch on both
    sides of the comparison."""
    bs = block_extractor_fwd(source.contiguous(), flow.contiguous(), k)
    bt = block_extractor_fwd(target.contiguous(), torch.zeros_like(flow), k)
    h = F.conv2d(torch.cat((bt, bs), 1), w0, b0, stride=k)
    h = F.leaky_relu(h, negative_slope)
    attn_ = F.softmax(F.conv2d(h, w1, b1), dim=1)
    attn = local_attn_reshape_fwd(attn_.contiguous(), k)
    res = F.avg_pool2d(attn * bs, k, k)
    return (attn_, res) if return_attn else res


# --------------------------------------------------------------------------- identity layer
def block_extractor_gather(source, flow, k):
    """Independent, differentiable formulation: explicit gather of the four taps."""
    B, C, Hs, Ws = source.shape
    _, _, Hf, Wf = flow.shape
    dev, dt = source.device, source.dtype
    y = torch.arange(k * Hf, device=dev)
    x = torch.arange(k * Wf, device=dev)
    yf, xf = y // k, x // k
    oy = (y % k - k // 2).to(dt)
    ox = (x % k - k // 2).to(dt)
    fx = flow[:, 0][:, yf][:, :, xf]  # (B, kHf, kWf)
    fy = flow[:, 1][:, yf][:, :, xf]
    dx = (fx + ox.view(1, 1, -1)) + xf.to(dt).view(1, 1, -1)
    dy = (fy + oy.view(1, -1, 1)) + yf.to(dt).view(1, -1, 1)
    x0, y0 = torch.floor(dx), torch.floor(dy)
    ax, ay = dx - x0, dy - y0
    xL = x0.long().clamp(0, Ws - 1)
    xR = (x0.long() + 1).clamp(0, Ws - 1)
    yT = y0.long().clamp(0, Hs - 1)
    yB = (y0.long() + 1).clamp(0, Hs - 1)
    flat = source.reshape(B, C, Hs * Ws)

    def tap(yy, xx):
        idx = (yy * Ws + xx).view(B, 1, -1).expand(B, C, -1)
        return flat.gather(2, idx).view(B, C, k * Hf, k * Wf)

    ax, ay = ax.unsqueeze(1), ay.unsqueeze(1)
    return ((1 - ax) * (1 - ay) * tap(yT, xL) + ax * (1 - ay) * tap(yT, xR)
            + (1 - ax) * ay * tap(yB, xL) + ax * ay * tap(yB, xR))


def block_extractor_grid_sample(source, flow, k):
    """Independent formulation through F.grid_sample(border, align_corners=True)."""
    B, C, Hs, Ws = source.shape
    _, _, Hf, Wf = flow.shape
    dt = source.dtype
    y = torch.arange(k * Hf)
    x = torch.arange(k * Wf)
    yf, xf = y // k, x // k
    gx = flow[:, 0][:, yf][:, :, xf] + (x % k - k // 2 + xf).to(dt).view(1, 1, -1)
    gy = flow[:, 1][:, yf][:, :, xf] + (y % k - k // 2 + yf).to(dt).view(1, -1, 1)
    grid = torch.stack((2 * gx / max(Ws - 1, 1) - 1, 2 * gy / max(Hs - 1, 1) - 1), -1)
    return F.grid_sample(source, grid, mode="bilinear", padding_mode="border", align_corners=True)


def resample2d_gather(input1, input2, k=2, dilation=1):
    """Independent, differentiable formulation of the Gaussian resampler (floor everywhere)."""
    _, C, Hi, Wi = input1.shape
    B, _, H, W = input2.shape
    dt = input1.dtype
    xs = torch.arange(W).to(dt).view(1, 1, W)
    ys = torch.arange(H).to(dt).view(1, H, 1)
    xf = xs + input2[:, 0]
    yf = ys + input2[:, 1]
    sig = input2[:, 2]
    x0, y0 = torch.floor(xf), torch.floor(yf)
    a, bta = xf - x0, yf - y0
    flat = input1.reshape(B, C, Hi * Wi)
    val = torch.zeros(B, C, H, W, dtype=dt)
    tot = torch.zeros(B, 1, H, W, dtype=dt)
    den = 2 * sig * sig

    def tap(yy, xx):
        idx = (yy * Wi + xx).view(B, 1, -1).expand(B, C, -1)
        return flat.gather(2, idx).view(B, C, H, W)

    for fy in range(k // 2):
        yT = (y0.long() - fy * dilation).clamp(0, Hi - 1)
        yB = (y0.long() + (fy + 1) * dilation).clamp(0, Hi - 1)
        wT = torch.exp(-(fy * dilation + bta) ** 2 / den)
        wB = torch.exp(-((1 + fy) * dilation - bta) ** 2 / den)
        for fx in range(k // 2):
            xL = (x0.long() - fx * dilation).clamp(0, Wi - 1)
            xR = (x0.long() + (fx + 1) * dilation).clamp(0, Wi - 1)
            wL = torch.exp(-(fx * dilation + a) ** 2 / den)
            wR = torch.exp(-((1 + fx) * dilation - a) ** 2 / den)
            for wy, yy in ((wT, yT), (wB, yB)):
                for wx, xx in ((wL, xL), (wR, xR)):
                    w = (wy * wx).unsqueeze(1)
                    val = val + w * tap(yy, xx)
                    tot = tot + w
    return val / tot


def aggregate_identity(attn_logits_softmaxed, block_source, k):
    """avg_pool(pixel_shuffle(a) * bs, k) == (1/k^2) sum_ij a_ij * bs_ij  (SURVEY 8c)."""
    return F.avg_pool2d(F.pixel_shuffle(attn_logits_softmaxed, k) * block_source, k, k)
