"""LocalAttnReshape on gfx950: (B, k*k, H, W) -> (B, 1, k*H, k*W) depth-to-space.

The k*k attention weights the FC head of ExtractorAttn predicts for every pixel are laid out as a
k x k tile per pixel so they can be multiplied with the extracted patches:

    tiled[b, 0, y*k + i, x*k + j] = inputs[b, i*k + j, y, x]          (== F.pixel_shuffle(inputs, k))

Public surface identical to the reference's model/networks/local_attn_reshape/local_attn_reshape.py
(`LocalAttnReshapeFunction.apply(inputs, kernel_size)` with gradients `(grad_inputs, None)`, :5-37;
`LocalAttnReshape().forward(inputs, kernel_size=3)`, :40-46).  Both directions are pure permutations
executed as gathers by libgfla_hip.so (bit-exact; the reference's backward uses atomics on what is a
bijection).
"""
from torch import nn
from torch.autograd import Function

from . import _lib

_FWD = "gfla_local_attn_reshape_fwd_"
_BWD = "gfla_local_attn_reshape_bwd_"


def _launch(entry, src, dst, B, H, W, k):
    """One permutation launch on src's device and current stream; empty tensors launch nothing."""
    if dst.numel():
        _lib.call(entry + _lib.suffix(src, "local_attn_reshape"), src, _lib.ptr(src), _lib.ptr(dst), B, H, W, k)
    return dst


class LocalAttnReshapeFunction(Function):
    """autograd wrapper; `kernel_size` is a plain int and receives no gradient."""

    @staticmethod
    def forward(ctx, inputs, kernel_size):
        k = int(kernel_size)
        assert inputs.is_contiguous()
        _lib.require_gpu(inputs)
        B, KK, H, W = inputs.shape
        assert KK == k * k  # as local_attn_reshape.py:13
        ctx.kernel_size = k
        ctx.in_shape = (B, KK, H, W)
        return _launch(_FWD, inputs, inputs.new_empty((B, 1, k * H, k * W)), B, H, W, k)

    @staticmethod
    def backward(ctx, grad_tiled):
        B, _, H, W = ctx.in_shape
        grad_tiled = grad_tiled.contiguous()
        # the inverse permutation writes every element once: no zero fill, no accumulation
        return _launch(_BWD, grad_tiled, grad_tiled.new_empty(ctx.in_shape), B, H, W, ctx.kernel_size), None


class LocalAttnReshape(nn.Module):
    """Parameter-free module; the tile size is an argument of forward(), as in the reference."""

    def __init__(self):
        super(LocalAttnReshape, self).__init__()

    def forward(self, inputs, kernel_size=3):
        return LocalAttnReshapeFunction.apply(inputs.contiguous(), kernel_size)
