"""Resample2d on gfx950: normalised Gaussian k x k flow warping with a per-pixel sigma.

Public surface identical to the reference's model/networks/resample2d_package/resample2d.py
(`Resample2dFunction.apply(input1, input2[B,3,H,W], kernel_size=2, dilation=1)` with gradients
`(g1, g2, None, None)`, :6-39; `Resample2d(kernel_size=2, dilation=1, sigma=5).forward(input1,
input2[B,2,H,W])`, which appends the constant sigma channel, :41-53).
"""
import torch
from torch import nn
from torch.autograd import Function

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
        B1, C, Hi, Wi = input1.shape
        B, three, H, W = input2.shape
        if three != 3:
            raise ValueError("resample2d: input2 must be (B,3,H,W) = (dx,dy,sigma)")
        if B1 != B:
            raise ValueError("resample2d: input1 batch %d != input2 batch %d" % (B1, B))
        if input1.dtype != input2.dtype:
            raise TypeError("resample2d: input1 is %s but input2 is %s" % (input1.dtype, input2.dtype))
        ctx.kernel_size, ctx.dilation = int(kernel_size), int(dilation)
        ctx.save_for_backward(input1, input2)
        warped = input1.new_empty((B, C, H, W))  # output takes b,h,w from input2 and d from input1 (:17-19)
        if warped.numel() == 0 or input1.numel() == 0:
            return warped.zero_()
        _lib.call("gfla_resample2d_fwd_" + _lib.suffix(input1, "resample2d"), input1,
                  _lib.ptr(input1), _lib.ptr(input2), _lib.ptr(warped),
                  B, C, Hi, Wi, H, W, ctx.kernel_size, ctx.dilation)
        return warped

    @staticmethod
    def backward(ctx, grad_warped):
        input1, input2 = ctx.saved_tensors
        want1, want2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        # d/d input1 is handed over UNINITIALISED (flag bit 1 below): the library overwrites it where every element has one
        # writer and zero-fills it itself where it has to accumulate with atomics
        g1 = torch.empty_like(input1) if want1 else None
        g2 = _lib.reduction_like(input2) if want2 else None  # float32 accumulator for bf16 storage
        grad_warped = grad_warped.contiguous()
        if (want1 or want2) and grad_warped.numel() > 0 and input1.numel() > 0:
            _, C, Hi, Wi = input1.shape
            B, _, H, W = input2.shape
            sfx = _lib.suffix(input1, "resample2d backward")
            tail = (B, C, Hi, Wi, H, W, ctx.kernel_size, ctx.dilation, 1 if TRUNC_COMPAT else 0)
            tail1 = tail[:-1] + (tail[-1] | 2,)  # bit 1: overwrite grad_in1
            # two independent kernels (scatter into input1 / reduction for (dx, dy, sigma)): one C-ABI
            # call each keeps them separately visible to profilers
            def run(i1, i2, gw, o1, o2, sfx_):
                entry_ = "gfla_resample2d_bwd_" + sfx_
                if o1 is not None:
                    if sfx_ == "f32":  # d/d input1 as a block-sparse product on the matrix cores when the shape allows
                        ws = _lib.scatter_workspace(i1, B, H, W, ctx.kernel_size * ctx.kernel_size)
                        _lib.call("gfla_resample2d_bwd_ws_f32", i1, _lib.ptr(i1), _lib.ptr(i2), _lib.ptr(gw), _lib.ptr(o1), None,
                                  _lib.ptr(ws), *tail1)
                    else:
                        _lib.call(entry_, i1, _lib.ptr(i1), _lib.ptr(i2), _lib.ptr(gw), _lib.ptr(o1), None, *tail1)
                if o2 is not None:
                    _lib.call(entry_, i1, _lib.ptr(i1), _lib.ptr(i2), _lib.ptr(gw), None, _lib.ptr(o2), *tail)

            try:
                run(input1, input2, grad_warped, g1, g2, sfx)
            except _lib.Unsupported:
                if sfx != "bf16":
                    raise
                # bfloat16 planes beyond the LDS budget: the bf16 backward exists for the planes-in-LDS kernels only; storage is
                # widened for this call (exact up-casts; d/d input1 rounded to bf16 once at the end) -- block_extractor.py
                i1, i2, gw = _lib.convert_many([input1, input2, grad_warped], torch.float32)
                g1_32 = torch.empty_like(i1) if want1 else None
                if g2 is not None:
                    g2.zero_()   # (a first, refused call may not have touched it; the float32 kernels accumulate into it)
                run(i1, i2, gw, g1_32, g2, "f32")
                g1 = None if g1_32 is None else _lib.convert_many([g1_32], torch.bfloat16)[0]
        elif g1 is not None:
            g1.zero_()  # nothing was launched (an empty input2 / gradient): d/d input1 is zero, not uninitialised memory
        if g2 is not None and g2.dtype != input2.dtype:
            g2 = g2.to(input2.dtype)
        return g1, g2, None, None


class Resample2d(nn.Module):

    def __init__(self, kernel_size=2, dilation=1, sigma=5):
        super(Resample2d, self).__init__()
        self.kernel_size = kernel_size
        self.dilation = dilation
        # a plain attribute, not a buffer, as in the reference (:47); materialised on first use so the
        # module can be constructed without a GPU
        self._sigma_value = float(sigma)
        self.sigma = None

    def forward(self, input1, input2):
        if self.sigma is None or self.sigma.device != input2.device:
            self.sigma = torch.tensor(self._sigma_value, dtype=torch.float, device=input2.device)
        B, _, H, W = input2.shape
        # (the reference concatenates the EXPANDED scalar, resample2d.py:51-52; torch.cat with a stride-0 operand leaves its fast
        # path: 49 us for this 1 MB tensor in the step's kernel trace, profiles/r5_final_bench_kernel_stats.csv -- so the plane
        # is materialised first: a 3 us fill, then the contiguous cat)
        sigma_plane = self.sigma.expand(B, 1, H, W).type(input2.dtype).contiguous()
        return Resample2dFunction.apply(input1.contiguous(), torch.cat((input2, sigma_plane), 1),
                                        self.kernel_size, self.dilation)
