"""Resample2d -- same surface as the reference's model/networks/resample2d_package/resample2d.py
(Resample2dFunction :6-39, Resample2d :41-53)."""
import torch
from torch.autograd import Function
from torch.nn.modules.module import Module

from . import _lib

# True reproduces the reference's int() truncation in the input1 gradient
# (resample2d_kernel.cu:137-138); False uses floor, the true gradient of the forward pass.
TRUNC_COMPAT = True


class Resample2dFunction(Function):

    @staticmethod
    def forward(ctx, input1, input2, kernel_size=2, dilation=1):
        assert input1.is_contiguous()
        assert input2.is_contiguous()
        _lib.require_gpu(input1, input2)
        if input2.size(1) != 3:
            raise ValueError("resample2d: input2 must be (B,3,H,W) = (dx,dy,sigma)")
        if input1.dtype != input2.dtype:
            raise TypeError("resample2d: input1 is %s but input2 is %s" % (input1.dtype, input2.dtype))
        ctx.save_for_backward(input1, input2)
        ctx.kernel_size = kernel_size
        ctx.dilation = dilation
        b1, d, hi, wi = input1.size()
        b, _, h, w = input2.size()
        if b1 != b:
            raise ValueError("resample2d: input1 batch %d != input2 batch %d" % (b1, b))
        output = input1.new_empty((b, d, h, w))
        if output.numel() == 0 or input1.numel() == 0:
            return output.zero_()
        _lib.call("gfla_resample2d_fwd_" + _lib.suffix(input1, "resample2d"), input1,
                  _lib.ptr(input1), _lib.ptr(input2), _lib.ptr(output),
                  b, d, hi, wi, h, w, int(kernel_size), int(dilation))
        return output

    @staticmethod
    def backward(ctx, grad_output):
        grad_output = grad_output.contiguous()
        input1, input2 = ctx.saved_tensors
        _, d, hi, wi = input1.size()
        b, _, h, w = input2.size()
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_input1 = torch.zeros_like(input1) if need1 else None
        grad_input2 = torch.zeros_like(input2) if need2 else None
        if (need1 or need2) and grad_output.numel() > 0 and input1.numel() > 0:
            # the two gradients are two independent kernels (scatter into input1 / reduction for
            # (dx, dy, sigma)); one C-ABI call each keeps them separately visible to profilers
            name = "gfla_resample2d_bwd_" + _lib.suffix(input1, "resample2d backward")
            trunc = 1 if TRUNC_COMPAT else 0
            if need1:
                _lib.call(name, input1, _lib.ptr(input1), _lib.ptr(input2), _lib.ptr(grad_output),
                          _lib.ptr(grad_input1), None, b, d, hi, wi, h, w, int(ctx.kernel_size), int(ctx.dilation), trunc)
            if need2:
                _lib.call(name, input1, _lib.ptr(input1), _lib.ptr(input2), _lib.ptr(grad_output),
                          None, _lib.ptr(grad_input2), b, d, hi, wi, h, w, int(ctx.kernel_size), int(ctx.dilation), trunc)
        return grad_input1, grad_input2, None, None


class Resample2d(Module):

    def __init__(self, kernel_size=2, dilation=1, sigma=5):
        super(Resample2d, self).__init__()
        self.kernel_size = kernel_size
        self.dilation = dilation
        # plain attribute, not a buffer, as in the reference (:47); created on first use so the
        # module can be constructed without a GPU.
        self._sigma_value = float(sigma)
        self.sigma = None

    def forward(self, input1, input2):
        input1_c = input1.contiguous()
        if self.sigma is None or self.sigma.device != input2.device:
            self.sigma = torch.tensor(self._sigma_value, dtype=torch.float, device=input2.device)
        sigma = self.sigma.expand(input2.size(0), 1, input2.size(2), input2.size(3)).type(input2.dtype)
        input2 = torch.cat((input2, sigma), 1)
        return Resample2dFunction.apply(input1_c, input2, self.kernel_size, self.dilation)
