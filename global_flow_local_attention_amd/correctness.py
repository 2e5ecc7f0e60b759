"""Sampling-correctness loss on the gfx950 ops.

Reference: `PerceptualCorrectness` (model/networks/external_function.py:223-319).  Per flow field it
  1. normalises the VGG features of source and target over channels and takes, for every target
     position, the best cosine similarity over ALL source positions -- through a materialised
     [b, N^2, N^2] `bmm` (:255-268);
  2. warps the source features with `Resample2d(4, 1, sigma=2)` (or `grid_sample`), takes the cosine
     similarity with the target features at the same position, and
  3. averages exp(-sample / (best + eps)), optionally under a mask (:270-277).

Step 1 is `max_cosine_similarity` below: one fp32 MFMA kernel in libgfla_hip.so that keeps the
similarity matrix in registers (gfla_max_cosine_fwd_f32).  Step 2 uses this package's Resample2d.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib
from .resample2d import Resample2d


class MaxCosineFunction(Function):
    """(source (B,C,Ns), target (B,C,Nt), eps) -> (best (B,Nt), index (B,Nt) int32).

    best[b,j] = max_i <source[b,:,i]/(|.|+eps), target[b,:,j]/(|.|+eps)>  (external_function.py:260-268).
    Backward re-evaluates the winning pairs only (B*C*Nt work): the max routes the gradient to one source
    position per target position, exactly as torch.max(dim=1) does in the reference.
    """

    @staticmethod
    def forward(ctx, source, target, eps):
        _lib.require_gpu(source, target)
        if source.dtype != torch.float32 or target.dtype != torch.float32:
            raise TypeError("max_cosine_similarity: float32 features only (got %s, %s)" % (source.dtype, target.dtype))
        assert source.is_contiguous() and target.is_contiguous()
        assert source.dim() == 3 and target.dim() == 3
        B, C, Ns = source.shape
        assert target.size(0) == B and target.size(1) == C
        Nt = target.size(2)
        best = source.new_empty(B, Nt)
        index = torch.empty(B, Nt, dtype=torch.int32, device=source.device)
        scratch = torch.empty(_lib.lib().gfla_max_cosine_workspace_bytes(B, Ns, Nt), dtype=torch.uint8,
                              device=source.device)
        _lib.call("gfla_max_cosine_fwd_f32", source, _lib.ptr(source), _lib.ptr(target), _lib.ptr(scratch),
                  _lib.ptr(best), _lib.ptr(index), B, C, Ns, Nt, float(eps))
        ctx.eps = eps
        ctx.save_for_backward(source, target, index)
        ctx.mark_non_differentiable(index)
        return best, index

    @staticmethod
    def backward(ctx, grad_best, _grad_index):
        source, target, index = ctx.saved_tensors
        need_s, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_s or need_t):
            return None, None, None
        with torch.enable_grad():
            s = source.detach().requires_grad_(need_s)
            t = target.detach().requires_grad_(need_t)
            winners = torch.gather(s, 2, index.long().unsqueeze(1).expand(-1, s.size(1), -1))
            winners = winners / (winners.norm(dim=1, keepdim=True) + ctx.eps)
            t_unit = t / (t.norm(dim=1, keepdim=True) + ctx.eps)
            best = (winners * t_unit).sum(1)
            wanted = [x for x, need in ((s, need_s), (t, need_t)) if need]
            grads = list(torch.autograd.grad(best, wanted, grad_best))
        return (grads.pop(0) if need_s else None), (grads.pop(0) if need_t else None), None


class CorrectnessMapFunction(Function):
    """(warped (B,C,N), target (B,C,N), best (B,N), eps) -> exp(-cosine_similarity(warped, target) / (best + eps)),
    external_function.py:275-276, as one pass forward and one backward."""

    COS_EPS = 1e-8  # F.cosine_similarity's default

    @staticmethod
    def forward(ctx, warped, target, best, eps):
        _lib.require_gpu(warped, target, best)
        for x in (warped, target, best):
            if x.dtype != torch.float32:
                raise TypeError("correctness map: float32 only (got %s)" % x.dtype)
            assert x.is_contiguous()
        B, C, N = warped.shape
        assert target.shape == warped.shape and best.shape == (B, N)
        loss_map = warped.new_empty(B, N)
        stats = warped.new_empty(B, N, 3)
        _lib.call("gfla_correctness_map_fwd_f32", warped, _lib.ptr(warped), _lib.ptr(target), _lib.ptr(best),
                  _lib.ptr(loss_map), _lib.ptr(stats), B, C, N, CorrectnessMapFunction.COS_EPS, float(eps))
        ctx.eps = eps
        ctx.save_for_backward(warped, target, best, stats, loss_map)
        return loss_map

    @staticmethod
    def backward(ctx, grad_map):
        warped, target, best, stats, loss_map = ctx.saved_tensors
        B, C, N = warped.shape
        need = ctx.needs_input_grad
        g_warped = torch.empty_like(warped) if need[0] else None
        g_target = torch.empty_like(target) if need[1] else None
        g_best = torch.empty_like(best) if need[2] else None
        if any(need[:3]):
            _lib.call("gfla_correctness_map_bwd_f32", warped, _lib.ptr(warped), _lib.ptr(target), _lib.ptr(best),
                      _lib.ptr(stats), _lib.ptr(loss_map), _lib.ptr(grad_map.contiguous()), _lib.ptr(g_warped),
                      _lib.ptr(g_target), _lib.ptr(g_best), B, C, N, CorrectnessMapFunction.COS_EPS, float(ctx.eps))
        return g_warped, g_target, g_best, None


def max_cosine_similarity(source, target, eps=1e-8, return_index=False):
    """Best cosine match over all source positions for every target position.

    source (B,C,...) and target (B,C,...) feature maps (any trailing spatial shape); returns (B, Nt)
    [and the int32 index of the winning source position]."""
    B, C = source.shape[:2]
    best, index = MaxCosineFunction.apply(source.reshape(B, C, -1).contiguous(),
                                          target.reshape(B, C, -1).contiguous(), eps)
    return (best, index) if return_index else best


class PerceptualCorrectness(nn.Module):
    """Same call surface as the reference class (external_function.py:223-319).

    The reference builds a pretrained torchvision VGG19 in its constructor; here the feature extractor
    is injected (`vgg`: callable image -> {layer name: feature map}), because neither torchvision nor
    its weights are part of this package.  `calculate_loss` works on `self.target_vgg` /
    `self.source_vgg` exactly as the reference's does, so it can also be driven with precomputed
    features.
    """

    def __init__(self, layer=['rel1_1', 'relu2_1', 'relu3_1', 'relu4_1'], vgg=None):
        super(PerceptualCorrectness, self).__init__()
        if isinstance(vgg, nn.Module):
            self.add_module('vgg', vgg)
        else:
            self.vgg = vgg
        self.layer = layer
        self.eps = 1e-8
        self.resample = Resample2d(4, 1, sigma=2)
        # grid_sample convention of `bilinear_warp`: the reference targets PyTorch 1.0.0 (README.md:93),
        # whose grid_sample had no align_corners argument and behaved as align_corners=True
        self.align_corners = True
        self.fused = True   # False: cosine_similarity / exp through torch ops, as the reference writes them

    def __call__(self, target, source, flow_list, used_layers, mask=None, use_bilinear_sampling=False):
        if self.vgg is None:
            raise RuntimeError("PerceptualCorrectness needs a feature extractor: pass vgg=... "
                               "(the reference's VGG19 requires torchvision weights)")
        used_layers = sorted(used_layers, reverse=True)
        self.target_vgg, self.source_vgg = self.vgg(target), self.vgg(source)
        total = 0
        for flow, which in zip(flow_list, used_layers):
            total = total + self.calculate_loss(flow, self.layer[which], mask, use_bilinear_sampling)
        return total

    def calculate_loss(self, flow, layer, mask=None, use_bilinear_sampling=False):
        target_feat = self.target_vgg[layer]
        source_feat = self.source_vgg[layer]
        b, c, h, w = target_feat.shape
        flow = F.interpolate(flow, [h, w])

        best = max_cosine_similarity(source_feat, target_feat, self.eps)              # :255-268
        if use_bilinear_sampling:
            warped = self.bilinear_warp(source_feat, flow)
        else:
            warped = self.resample(source_feat, flow).view(b, c, -1)                  # :273
        if self.fused and warped.dtype == torch.float32:
            loss_map = CorrectnessMapFunction.apply(warped.contiguous(), target_feat.reshape(b, c, -1).contiguous(),
                                                    best, self.eps)                   # :275-276
        else:
            sampled = F.cosine_similarity(warped, target_feat.view(b, c, -1))
            loss_map = torch.exp(-sampled / (best + self.eps))
        floor = torch.exp(torch.tensor(-1.0)).type_as(loss_map)
        if mask is None:
            return torch.mean(loss_map) - floor
        mask = F.interpolate(mask, size=(h, w)).view(-1, h * w)
        return torch.sum(mask * (loss_map - floor)) / (torch.sum(mask) + self.eps)

    def bilinear_warp(self, source, flow):
        """grid_sample alternative of the reference (:308-318), same normalisation of the flow."""
        b, c, h, w = source.shape
        xs = torch.arange(w, device=source.device).view(1, -1).expand(h, -1).type_as(source) / (w - 1)
        ys = torch.arange(h, device=source.device).view(-1, 1).expand(-1, w).type_as(source) / (h - 1)
        grid = 2 * torch.stack([xs, ys], dim=0).unsqueeze(0).expand(b, -1, -1, -1) - 1
        scale = torch.tensor([w, h], device=flow.device).view(1, 2, 1, 1).type_as(flow)
        grid = (grid + 2 * flow / scale).permute(0, 2, 3, 1)
        return F.grid_sample(source, grid, align_corners=self.align_corners).view(b, c, -1)
