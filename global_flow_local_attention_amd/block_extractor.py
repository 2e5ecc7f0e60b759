"""BlockExtractor on gfx950.

Public surface identical to the reference's model/networks/block_extractor/block_extractor.py
(`BlockExtractorFunction.apply(source, flow_field, kernel_size)` returning gradients
`(grad_source, grad_flow_field, None)`, :5-42; `BlockExtractor(kernel_size=3).forward(source,
flow_field)`, :45-54) so the reference's network code consumes it unchanged; the body is a ctypes
call into libgfla_hip.so instead of the `block_extractor_cuda` pybind module.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import _lib

_ENTRY_FWD = "gfla_block_extractor_fwd_"
_ENTRY_BWD = "gfla_block_extractor_bwd_"


def _geometry(source, flow_field):
    """(B, C, Hs, Ws, Hf, Wf) after the checks the reference makes (contiguity, two flow channels)
    plus the ones it leaves out (matching batch and dtype; :15 is commented out there)."""
    assert source.is_contiguous()
    assert flow_field.is_contiguous()
    _lib.require_gpu(source, flow_field)
    B, C, Hs, Ws = source.shape
    Bf, two, Hf, Wf = flow_field.shape
    assert two == 2
    if Bf != B:
        raise ValueError("block_extractor: source batch %d != flow batch %d" % (B, Bf))
    if flow_field.dtype != source.dtype:
        raise TypeError("block_extractor: source is %s but flow_field is %s" % (source.dtype, flow_field.dtype))
    return B, C, Hs, Ws, Hf, Wf


class BlockExtractorFunction(Function):

    @staticmethod
    def forward(ctx, source, flow_field, kernel_size):
        B, C, Hs, Ws, Hf, Wf = _geometry(source, flow_field)
        k = int(kernel_size)
        ctx.kernel_size = k
        ctx.save_for_backward(source, flow_field)
        # every element is written by the kernel: no zero fill (the reference zero-fills, :21)
        patches = flow_field.new_empty((B, C, k * Hf, k * Wf))
        if patches.numel() == 0 or source.numel() == 0:  # empty batch / channels: nothing to launch
            return patches.zero_()
        _lib.call(_ENTRY_FWD + _lib.suffix(source, "block_extractor"), source,
                  _lib.ptr(source), _lib.ptr(flow_field), _lib.ptr(patches), B, C, Hs, Ws, Hf, Wf, k)
        return patches

    @staticmethod
    def backward(ctx, grad_patches):
        source, flow_field = ctx.saved_tensors
        want_source, want_flow = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_source = torch.zeros_like(source) if want_source else None
        g_flow = _lib.reduction_like(flow_field) if want_flow else None  # float32 accumulator for bf16 storage
        grad_patches = grad_patches.contiguous()  # the reference drops this result (:32-33)
        if (want_source or want_flow) and grad_patches.numel() > 0 and source.numel() > 0:
            B, C, Hs, Ws = source.shape
            Hf, Wf = flow_field.shape[2:]
            try:
                _lib.call(_ENTRY_BWD + _lib.suffix(source, "block_extractor backward"), source,
                          _lib.ptr(source), _lib.ptr(flow_field), _lib.ptr(grad_patches),
                          _lib.ptr(g_source), _lib.ptr(g_flow), B, C, Hs, Ws, Hf, Wf, ctx.kernel_size)
            except _lib.Unsupported:
                if source.dtype != torch.bfloat16:
                    raise
                # bfloat16 planes beyond the LDS budget (e.g. one (1,64,256,176) map): the bf16 backward exists for the
                # planes-in-LDS kernels only -- the tile kernels accumulate into float32 / float64 windows and flush with float
                # atomics.  Storage is widened for this call: the float32 kernels run on exact up-casts of the bf16 values, the
                # source gradient is rounded to bf16 once at the end (one rounding per element, as the bf16 kernels do).
                s32, f32_, gp32 = _lib.convert_many([source, flow_field, grad_patches], torch.float32)
                gs32 = torch.zeros_like(s32) if want_source else None
                _lib.call(_ENTRY_BWD + "f32", s32, _lib.ptr(s32), _lib.ptr(f32_), _lib.ptr(gp32), _lib.ptr(gs32),
                          _lib.ptr(g_flow), B, C, Hs, Ws, Hf, Wf, ctx.kernel_size)
                g_source = None if gs32 is None else _lib.convert_many([gs32], torch.bfloat16)[0]
        if g_flow is not None and g_flow.dtype != flow_field.dtype:
            g_flow = g_flow.to(flow_field.dtype)
        return g_source, g_flow, None


class BlockExtractor(nn.Module):
    """k x k bilinear patch around (x, y) + flow for every flow pixel: (B,C,Hs,Ws), (B,2,Hf,Wf) ->
    (B,C,k*Hf,k*Wf)."""

    def __init__(self, kernel_size=3):
        super(BlockExtractor, self).__init__()
        self.kernel_size = kernel_size

    def forward(self, source, flow_field):
        return BlockExtractorFunction.apply(source.contiguous(), flow_field.contiguous(), self.kernel_size)
