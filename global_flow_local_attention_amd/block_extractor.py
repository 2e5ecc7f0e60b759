"""BlockExtractor -- same surface as the reference's model/networks/block_extractor/block_extractor.py
(BlockExtractorFunction :5-42, BlockExtractor :45-54), backed by the gfx950 kernels."""
import torch
from torch.autograd import Function
from torch.nn.modules.module import Module

from . import _lib


class BlockExtractorFunction(Function):

    @staticmethod
    def forward(ctx, source, flow_field, kernel_size):
        assert source.is_contiguous()
        assert flow_field.is_contiguous()
        _lib.require_gpu(source, flow_field)
        bs, ds, hs, ws = source.size()
        bf, df, hf, wf = flow_field.size()
        assert df == 2
        if bf != bs:
            raise ValueError("block_extractor: source batch %d != flow batch %d" % (bs, bf))
        if flow_field.dtype != source.dtype:
            raise TypeError("block_extractor: source is %s but flow_field is %s" % (source.dtype, flow_field.dtype))
        ctx.save_for_backward(source, flow_field)
        ctx.kernel_size = kernel_size
        # the kernel writes every element, so no zero fill (the reference zero-fills, :21)
        output = flow_field.new_empty((bs, ds, kernel_size * hf, kernel_size * wf))
        if output.numel() == 0 or source.numel() == 0:  # empty batch / channels: nothing to launch
            return output.zero_()
        _lib.call("gfla_block_extractor_fwd_" + _lib.suffix(source, "block_extractor"), source,
                  _lib.ptr(source), _lib.ptr(flow_field), _lib.ptr(output),
                  bs, ds, hs, ws, hf, wf, int(kernel_size))
        return output

    @staticmethod
    def backward(ctx, grad_output):
        grad_output = grad_output.contiguous()  # the reference drops this result (:32-33)
        source, flow_field = ctx.saved_tensors
        bs, ds, hs, ws = source.size()
        _, _, hf, wf = flow_field.size()
        need_src, need_flow = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_source = torch.zeros_like(source) if need_src else None
        grad_flow_field = torch.zeros_like(flow_field) if need_flow else None
        if (need_src or need_flow) and grad_output.numel() > 0 and source.numel() > 0:
            _lib.call("gfla_block_extractor_bwd_" + _lib.suffix(source, "block_extractor backward"), source,
                      _lib.ptr(source), _lib.ptr(flow_field), _lib.ptr(grad_output),
                      _lib.ptr(grad_source), _lib.ptr(grad_flow_field),
                      bs, ds, hs, ws, hf, wf, int(ctx.kernel_size))
        return grad_source, grad_flow_field, None


class BlockExtractor(Module):
    def __init__(self, kernel_size=3):
        super(BlockExtractor, self).__init__()
        self.kernel_size = kernel_size

    def forward(self, source, flow_field):
        source_c = source.contiguous()
        flow_field_c = flow_field.contiguous()
        return BlockExtractorFunction.apply(source_c, flow_field_c, self.kernel_size)
