"""TEST INFRASTRUCTURE ONLY -- the reference's module surface on the CPU, backed by the oracle.

The reference has no CPU path (block_extractor.py:23-24 raises), so "the reference on the host
cores" is the reference's own composition (base_function.py:790-818, resample2d.py:41-53) with the
oracle's literal kernels underneath.  bench.py's cpu_baseline leg and the tests use these; the
product never does.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import cpu_oracle as O


class _BlockExtractorCPU(Function):
    @staticmethod
    def forward(ctx, source, flow, k):
        ctx.save_for_backward(source, flow)
        ctx.k = k
        return O.block_extractor_fwd(source.contiguous(), flow.contiguous(), k)

    @staticmethod
    def backward(ctx, g):
        source, flow = ctx.saved_tensors
        gs, gf = O.block_extractor_bwd(source.contiguous(), flow.contiguous(), g.contiguous(), ctx.k)
        return gs, gf, None


class _LocalAttnReshapeCPU(Function):
    @staticmethod
    def forward(ctx, x, k):
        ctx.k = k
        return O.local_attn_reshape_fwd(x.contiguous(), k)

    @staticmethod
    def backward(ctx, g):
        return O.local_attn_reshape_bwd(g.contiguous(), ctx.k), None


class _Resample2dCPU(Function):
    @staticmethod
    def forward(ctx, i1, i2, k, d):
        ctx.save_for_backward(i1, i2)
        ctx.k, ctx.d = k, d
        return O.resample2d_fwd(i1.contiguous(), i2.contiguous(), k, d)

    @staticmethod
    def backward(ctx, g):
        i1, i2 = ctx.saved_tensors
        g1, g2 = O.resample2d_bwd(i1.contiguous(), i2.contiguous(), g.contiguous(), ctx.k, ctx.d, True)
        return g1, g2, None, None


class Resample2dCPU(nn.Module):
    def __init__(self, kernel_size=2, dilation=1, sigma=5):
        super().__init__()
        self.kernel_size, self.dilation, self.sigma = kernel_size, dilation, float(sigma)

    def forward(self, input1, input2):
        sig = torch.full_like(input2[:, :1], self.sigma)
        return _Resample2dCPU.apply(input1, torch.cat((input2, sig), 1), self.kernel_size, self.dilation)


class ExtractorAttnCPU(nn.Module):
    """base_function.py:790-810, op by op."""

    def __init__(self, feature_nc, kernel_size=4, nonlinearity=nn.LeakyReLU(), softmax=None):
        super().__init__()
        self.kernel_size = kernel_size
        softmax = nonlinearity if softmax is None else nn.Softmax(dim=1)
        self.fully_connect_layer = nn.Sequential(
            nn.Conv2d(2 * feature_nc, 128, kernel_size=kernel_size, stride=kernel_size, padding=0),
            nonlinearity,
            nn.Conv2d(128, kernel_size * kernel_size, kernel_size=1, stride=1, padding=0),
            softmax,)

    def forward(self, source, target, flow_field):
        k = self.kernel_size
        block_source = _BlockExtractorCPU.apply(source, flow_field, k)
        block_target = _BlockExtractorCPU.apply(target, torch.zeros_like(flow_field), k)
        attn = self.fully_connect_layer(torch.cat((block_target, block_source), 1))
        attn = _LocalAttnReshapeCPU.apply(attn, k)
        return F.avg_pool2d(attn * block_source, k, k)


def max_cosine_cpu(source, target, eps=1e-8):
    """external_function.py:255-268 on host tensors: source (B,C,Ns), target (B,C,Nt) ->
    (best (B,Nt), index (B,Nt)).  Materialises the [B,Ns,Nt] similarity matrix like the reference."""
    source_all = source.transpose(1, 2)                                        # [b Ns C]
    source_norm = source_all / (source_all.norm(dim=2, keepdim=True) + eps)
    target_norm = target / (target.norm(dim=1, keepdim=True) + eps)
    correction = torch.bmm(source_norm, target_norm)                           # [b Ns Nt]
    return torch.max(correction, dim=1)


class PerceptualCorrectnessCPU(nn.Module):
    """external_function.py:246-279 (calculate_loss) op by op on the host, with the CPU Resample2d."""

    def __init__(self, vgg=None, layer=('rel1_1', 'relu2_1', 'relu3_1', 'relu4_1')):
        super().__init__()
        self.eps = 1e-8
        self.resample = Resample2dCPU(4, 1, sigma=2)
        self.target_vgg, self.source_vgg = {}, {}
        self.vgg, self.layer = vgg, list(layer)

    def __call__(self, target, source, flow_list, used_layers, mask=None):
        """external_function.py:235-243 with an injected feature extractor."""
        used_layers = sorted(used_layers, reverse=True)
        self.target_vgg, self.source_vgg = self.vgg(target), self.vgg(source)
        loss = 0
        for i in range(len(flow_list)):
            loss = loss + self.calculate_loss(flow_list[i], self.layer[used_layers[i]], mask)
        return loss

    def calculate_loss(self, flow, layer, mask=None):
        target_vgg, source_vgg = self.target_vgg[layer], self.source_vgg[layer]
        b, c, h, w = target_vgg.shape
        flow = F.interpolate(flow, [h, w])
        target_all = target_vgg.view(b, c, -1)
        correction_max, _ = max_cosine_cpu(source_vgg.view(b, c, -1), target_all, self.eps)
        input_sample = self.resample(source_vgg, flow).view(b, c, -1)
        correction_sample = F.cosine_similarity(input_sample, target_all)
        loss_map = torch.exp(-correction_sample / (correction_max + self.eps))
        e1 = torch.exp(torch.tensor(-1.0)).type_as(loss_map)
        if mask is None:
            return torch.mean(loss_map) - e1
        mask = F.interpolate(mask, size=(h, w)).view(-1, h * w)
        return torch.sum(mask * (loss_map - e1)) / (torch.sum(mask) + self.eps)


class AffineRegularizationLossOpByOp(nn.Module):
    """external_function.py:31-77 op by op: conv2d with the (k^2,1,k,k) projector -> LocalAttnReshape ->
    BlockExtractor(grid, constant flow k//2) -> multiply -> avg_pool2d -> mean * k^2.  `extractor(source, flow)` /
    `reshape(x, k)` default to the CPU oracle's ops; the GPU tests pass the library's modules instead to exercise them
    in this composition (Hs != Hf, C = 1)."""

    def __init__(self, kz, extractor=None, reshape=None):
        super().__init__()
        import numpy as np
        self.kz = kz
        self.extractor = extractor or (lambda s, f: _BlockExtractorCPU.apply(s, f, kz))
        self.reshape = reshape or (lambda x, k: _LocalAttnReshapeCPU.apply(x, k))
        temp = np.arange(kz)
        A = np.ones([kz * kz, 3])
        A[:, 0] = temp.repeat(kz)
        A[:, 1] = temp.repeat(kz).reshape((kz, kz)).transpose().reshape(kz ** 2)
        AH = A.transpose()
        k = np.dot(A, np.dot(np.linalg.inv(np.dot(AH, A)), AH)) - np.identity(kz ** 2)
        k = np.dot(k.transpose(), k)
        self.kernel = torch.from_numpy(k).unsqueeze(1).view(kz ** 2, kz, kz).unsqueeze(1)

    def forward(self, flow_fields):
        b, c, h, w = flow_fields.size()
        x = torch.arange(w).view(1, -1).expand(h, -1).type_as(flow_fields).float()
        y = torch.arange(h).view(-1, 1).expand(-1, w).type_as(flow_fields).float()
        grid = flow_fields + torch.stack([x, y], dim=0).unsqueeze(0).expand(b, -1, -1, -1)
        weights = self.kernel.type_as(flow_fields)
        return self._axis(grid[:, 0:1], weights) + self._axis(grid[:, 1:2], weights)

    def _axis(self, grid, weights):
        results = F.conv2d(grid, weights)
        b, c, h, w = results.size()
        kernels_new = self.reshape(results.contiguous(), self.kz)
        f = torch.zeros(b, 2, h, w).type_as(kernels_new) + float(int(self.kz / 2))
        grid_h = self.extractor(grid.contiguous(), f)
        result = F.avg_pool2d(grid_h * kernels_new, self.kz, self.kz)
        return torch.mean(result) * self.kz ** 2
