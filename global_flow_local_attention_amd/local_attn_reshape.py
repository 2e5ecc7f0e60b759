"""LocalAttnReshape -- same surface as the reference's
model/networks/local_attn_reshape/local_attn_reshape.py (Function :5-37, Module :40-46)."""
from torch.autograd import Function
from torch.nn.modules.module import Module

from . import _lib


class LocalAttnReshapeFunction(Function):

    @staticmethod
    def forward(ctx, inputs, kernel_size):
        assert inputs.is_contiguous()
        _lib.require_gpu(inputs)
        bs, ds, hs, ws = inputs.size()
        assert ds == kernel_size * kernel_size
        ctx.kernel_size = kernel_size
        ctx.in_shape = (bs, ds, hs, ws)
        output = inputs.new_empty((bs, 1, kernel_size * hs, kernel_size * ws))
        if output.numel() == 0:
            return output
        _lib.call("gfla_local_attn_reshape_fwd_" + _lib.suffix(inputs, "local_attn_reshape"), inputs,
                  _lib.ptr(inputs), _lib.ptr(output), bs, hs, ws, int(kernel_size))
        return output

    @staticmethod
    def backward(ctx, grad_output):
        grad_output = grad_output.contiguous()
        bs, ds, hs, ws = ctx.in_shape
        grad_inputs = grad_output.new_empty(ctx.in_shape)  # fully overwritten (a bijection)
        if grad_inputs.numel() == 0:
            return grad_inputs, None
        _lib.call("gfla_local_attn_reshape_bwd_" + _lib.suffix(grad_output, "local_attn_reshape"), grad_output,
                  _lib.ptr(grad_output), _lib.ptr(grad_inputs), bs, hs, ws, int(ctx.kernel_size))
        return grad_inputs, None


class LocalAttnReshape(Module):
    def __init__(self):
        super(LocalAttnReshape, self).__init__()

    def forward(self, inputs, kernel_size=3):
        inputs_c = inputs.contiguous()
        return LocalAttnReshapeFunction.apply(inputs_c, kernel_size)
