"""fully_connect_layer of ExtractorAttn on the matrix cores (reference: model/networks/base_function.py:799-807).

`FcMfmaFunction` maps (source, target, flow, conv0.weight, conv0.bias, conv1.weight, conv1.bias) to the attention
logits (B, k*k, H, W) -- what the reference computes as
    fully_connect_layer[:3](cat(extractor(target, 0), extractor(source, flow)))
-- through gfla_fc_forward_f32 / gfla_fc_backward_f32 (csrc/fc_block.hip): no block tensor, no library GEMM or
convolution.  `mode` picks the arithmetic of the contraction (include/gfla_hip.h):
  4  float32 throughout, Winograd-domain convolutions and weight gradient (csrc/fc_wino.hip: F(2x2,5x5) / F(4x4,3x3),
     2.78x / 4x fewer multiplies); errors against float64 at the bench shapes 1e-6 .. 9e-6, held to the same test bars as
     mode 0;
  0  float32, direct convolution: a k-ordered fma chain per output (what mode 4 falls back to for maps its tiles do not fit);
  5  float32 tensors; every operand of a product as TWO f16 terms (hi + lo = the value to 2^-24 after a power-of-two scaling
     from the tensor's max |x|) on the f16 matrix cores, f32 accumulation -- THE DEFAULT (round 6).  Which kernel runs what is
     decided per convolution by measurement (csrc/fc_block.hip: fc_hyb): the k = 5 convolutions and every data gradient on the
     direct kernels (three cross products, mode 2's arithmetic) reading the float32 maps in place; the k = 3 forward and the
     k = 5 weight gradient in the Winograd domain (all four cross products: csrc/fc_wino16.hip, fc_wino.hip); the k = 3 weight
     gradient is mode 4's kernel.  Same measured error as mode 4, same test bars;
  3 / 2  operands split into three / two f16 terms, f32 accumulation (labelled experiments);  1  one f16 term (bf16 path).
"""
import ctypes

import torch
from torch.autograd import Function

from . import _lib

MODES = (0, 1, 2, 3, 4, 5)
# float32 is what the reference computes this layer in (base_function.py:799-810): a module without an explicit `fc_mode`
# gets float32-grade arithmetic -- two f16 terms per operand on the f16 matrix cores (5: every operand represented to
# 2^-24, f32 accumulation; measured error = mode 4's), the float32 Winograd kernels (4) or the
# float32 direct kernels (0) where the shape does not fit the faster one; 2 / 3 are labelled experiments (lower / other
# precision trade-offs), 1 belongs to the bf16-feature path
DEFAULT_MODE = 5
MODE_NAMES = {0: "f32 MFMA, direct convolution", 1: "one f16 term per operand (exact for bf16 values), f32 accumulate",
              2: "two f16 terms per operand, f32 accumulate", 3: "three f16 terms per operand, f32 accumulate",
              4: "f32 MFMA, Winograd-domain convolutions F(2x2,5x5) / F(4x4,3x3)",
              5: "two f16 terms per operand on f16 MFMA, f32 accumulate: direct kernels (k5 convolutions, data gradients) + "
                 "Winograd domain F(2x2,5x5) / F(4x4,3x3) (k3 forward, k5 weight gradient)"}


def supported(C, H, W, k, mode=DEFAULT_MODE):
    return bool(_lib.lib().gfla_fc_supported(int(C), int(H), int(W), int(k), int(mode)))


def resolve_mode(C, H, W, k, mode=None):
    """The arithmetic mode a call will run in: `mode` (DEFAULT_MODE when None) if the kernels take the shape; the float32
    direct kernels (0) when the float32 Winograd kernels (4) do not; None when nothing does."""
    mode = DEFAULT_MODE if mode is None else int(mode)
    if mode not in MODES:
        return None
    if supported(C, H, W, k, mode):
        return mode
    if mode == 5 and supported(C, H, W, k, 4):
        return 4
    if mode in (4, 5) and supported(C, H, W, k, 0):
        return 0
    return None


def workspace_bytes(B, C, H, W, k, mode, which):
    n = _lib.lib().gfla_fc_workspace_bytes(int(B), int(C), int(H), int(W), int(k), int(mode), int(which))
    if n < 0:
        raise ValueError("fc_mfma: unsupported shape B=%d C=%d %dx%d k=%d mode=%d" % (B, C, H, W, k, mode))
    return n


def geometry(H, W, k, is_source):
    out = (ctypes.c_int64 * 13)()
    rc = _lib.lib().gfla_fc_geometry(int(H), int(W), int(k), 1 if is_source else 0, ctypes.cast(out, ctypes.c_void_p))
    if rc != 0:
        raise ValueError("gfla_fc_geometry: status %d" % rc)
    names = ("Hp", "Wp", "Ho", "Wo", "pad_t", "pad_l", "M", "Md", "lead", "Sx", "Sz", "Mg", "Mdg")
    return dict(zip(names, (int(v) for v in out)))


def _check(source, target, flow, w0, w1, k):
    _lib.require_gpu(source, target, flow, w0, w1)
    B, C, H, W = source.shape
    if tuple(target.shape) != (B, C, H, W) or tuple(flow.shape) != (B, 2, H, W):
        raise ValueError("fc_mfma: source %s, target %s and flow %s must share B, C (features) and H, W" %
                         (tuple(source.shape), tuple(target.shape), tuple(flow.shape)))
    if tuple(w0.shape) != (128, 2 * C, k, k) or tuple(w1.shape[:2]) != (k * k, 128):
        raise ValueError("fc_mfma: weights %s / %s do not belong to an ExtractorAttn(%d, %d)" %
                         (tuple(w0.shape), tuple(w1.shape), C, k))
    for t in (source, target, flow, w0, w1):
        if t.dtype != torch.float32:
            raise TypeError("fc_mfma: float32 only (got %s)" % t.dtype)


class FcMfmaFunction(Function):
    @staticmethod
    def forward(ctx, source, target, flow, w0, b0, w1, b1, kernel_size, slope, mode):
        k, mode = int(kernel_size), int(mode)
        _check(source, target, flow, w0, w1, k)
        source, target, flow = source.contiguous(), target.contiguous(), flow.contiguous()
        w0c, w1c = w0.contiguous(), w1.reshape(k * k, 128).contiguous()
        b0c = None if b0 is None else b0.contiguous()
        b1c = None if b1 is None else b1.contiguous()
        B, C, H, W = source.shape
        ws = torch.empty(workspace_bytes(B, C, H, W, k, mode, 0), dtype=torch.uint8, device=source.device)
        logits = source.new_empty((B, k * k, H, W))
        _lib.call("gfla_fc_forward_f32", source, _lib.ptr(source), _lib.ptr(target), _lib.ptr(flow), _lib.ptr(w0c),
                  _lib.ptr(b0c), _lib.ptr(w1c), _lib.ptr(b1c), _lib.ptr(ws), _lib.ptr(logits), B, C, H, W, k,
                  float(slope), mode)
        ctx.save_for_backward(flow, w1c, ws)
        ctx.dims = (B, C, H, W, k, float(slope), mode)
        ctx.w_shapes = (w0.shape, w1.shape, b0 is not None, b1 is not None)
        return logits

    @staticmethod
    def backward(ctx, g_logits):
        flow, w1c, ws = ctx.saved_tensors
        B, C, H, W, k, slope, mode = ctx.dims
        w0_shape, w1_shape, has_b0, has_b1 = ctx.w_shapes
        need = ctx.needs_input_grad
        g_logits = g_logits.contiguous()
        dev = flow.device

        def out(shape, wanted):
            return torch.empty(shape, dtype=torch.float32, device=dev) if wanted else None

        g_source, g_target = out((B, C, H, W), need[0]), out((B, C, H, W), need[1])
        g_flow = out((B, 2, H, W), need[2])
        g_w0 = out(w0_shape, need[3])
        g_b0 = out((128,), need[4] and has_b0)
        g_w1 = out(w1_shape, need[5])
        g_b1 = out((k * k,), need[6] and has_b1)
        scratch = torch.empty(workspace_bytes(B, C, H, W, k, mode, 1), dtype=torch.uint8, device=dev)
        _lib.call("gfla_fc_backward_f32", flow, _lib.ptr(ws), _lib.ptr(flow), _lib.ptr(w1c), _lib.ptr(g_logits),
                  _lib.ptr(scratch), _lib.ptr(g_source), _lib.ptr(g_target), _lib.ptr(g_flow), _lib.ptr(g_w0),
                  _lib.ptr(g_b0), _lib.ptr(g_w1), _lib.ptr(g_b1), B, C, H, W, k, slope, mode, 0)
        return g_source, g_target, g_flow, g_w0, g_b0, g_w1, g_b1, None, None, None
