"""Loss-side consumers of the hot path, restated so they need fewer passes.

`AffineRegularizationLoss` (reference: model/networks/external_function.py:31-77) runs, per flow field
and per axis, conv2d(grid, K) -> LocalAttnReshape -> BlockExtractor(grid, const flow k//2) -> multiply
-> avg_pool2d -> mean * k^2.  With u = the k x k patch of the sampling grid at a valid position,
conv2d gives (M u), the extractor at the constant integer flow k//2 returns exactly u (bilinear
weights 1/0), and avg_pool of the product is u.(M u)/k^2.  So the whole chain is

        loss_axis = mean over (b, valid positions) of  u^T M u ,      M = K^T K  (k^2 x k^2, fixed)

one unfold + one small GEMM + one reduction, with no custom-op launches at all.  The reference's op-by-op
composition lives in oracle/cpu_modules.py (AffineRegularizationLossOpByOp, test infrastructure); goldens produced by
the reference's own class pin both (tests/golden/make_affine_golden.py).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def affine_projector(kz):
    """M = K^T K with K = A (A^T A)^-1 A^T - I, A = [x, y, 1] over the kz x kz patch
    (external_function.py:41-47)."""
    temp = np.arange(kz)
    A = np.ones([kz * kz, 3])
    A[:, 0] = temp.repeat(kz)
    A[:, 1] = temp.repeat(kz).reshape((kz, kz)).transpose().reshape(kz ** 2)
    AH = A.transpose()
    k = np.dot(A, np.dot(np.linalg.inv(np.dot(AH, A)), AH)) - np.identity(kz ** 2)
    return torch.from_numpy(np.dot(k.transpose(), k))


class AffineRegularizationLoss(nn.Module):
    """Same constructor/call as the reference (external_function.py:31-77)."""

    def __init__(self, kz):
        super(AffineRegularizationLoss, self).__init__()
        self.kz = kz
        self.kernel = affine_projector(kz).view(kz ** 2, kz ** 2)

    def __call__(self, flow_fields):
        grid = self.flow2grid(flow_fields)
        weights = self.kernel.type_as(flow_fields)
        loss_x = self.calculate_loss(grid[:, 0:1], weights)
        loss_y = self.calculate_loss(grid[:, 1:2], weights)
        return loss_x + loss_y

    def calculate_loss(self, grid, weights):
        u = F.unfold(grid, self.kz)          # (B, kz^2, L): the valid kz x kz patches
        mu = torch.matmul(weights, u)        # M u
        return (u * mu).sum(1).mean()

    def flow2grid(self, flow_field):
        b, c, h, w = flow_field.size()
        x = torch.arange(w).view(1, -1).expand(h, -1).type_as(flow_field).float()
        y = torch.arange(h).view(-1, 1).expand(-1, w).type_as(flow_field).float()
        grid = torch.stack([x, y], dim=0).unsqueeze(0).expand(b, -1, -1, -1)
        return flow_field + grid


class MultiAffineRegularizationLoss(nn.Module):
    """external_function.py:12-27: one AffineRegularizationLoss per attention layer."""

    def __init__(self, kz_dic):
        super(MultiAffineRegularizationLoss, self).__init__()
        self.kz_dic = kz_dic
        self.method_dic = {}
        for key in kz_dic:
            self.method_dic[key] = AffineRegularizationLoss(kz_dic[key])
        self.layers = sorted(kz_dic, reverse=True)

    def __call__(self, flow_fields):
        loss = 0
        for i in range(len(flow_fields)):
            loss += self.method_dic[self.layers[i]](flow_fields[i])
        return loss
