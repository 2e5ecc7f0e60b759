#!/usr/bin/env python3
"""How exact are MIOpen's fp32 convolutions on this GPU?  (context for the ExtractorAttn FC tolerances)"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
dev = "cuda:0"
for (B, C, H, W, k, stride) in [(2, 32, 36, 30, 3, 3), (32, 256, 320, 220, 5, 5), (32, 128, 64, 44, 5, 1), (32, 512, 96, 66, 3, 3)]:
    x = torch.randn(B, C, H, W)
    w = torch.randn(128, C, k, k) * 0.05
    for tf32 in (True, False):
        torch.backends.cudnn.allow_tf32 = tf32
        xd, wd = x.to(dev).requires_grad_(), w.to(dev).requires_grad_()
        y = F.conv2d(xd, wd, stride=stride)
        g = torch.randn_like(y)
        y.backward(g)
        if B <= 2:
            x64, w64 = x.double().requires_grad_(), w.double().requires_grad_()
            y64 = F.conv2d(x64, w64, stride=stride)
            y64.backward(g.cpu().double())
            rel = lambda a, b: ((a.cpu().double() - b).abs().max() / b.abs().max()).item()
            print("B%d C%d %dx%d k%d s%d allow_tf32=%s  fwd rel %.2e  dgrad rel %.2e  wgrad rel %.2e" %
                  (B, C, H, W, k, stride, tf32, rel(y.detach(), y64.detach()), rel(xd.grad, x64.grad), rel(wd.grad, w64.grad)))
        else:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            xd.grad = wd.grad = None
            e0.record()
            for _ in range(5):
                y = F.conv2d(xd, wd, stride=stride)
            e1.record(); torch.cuda.synchronize()
            tf = e0.elapsed_time(e1) / 5
            e0.record()
            for _ in range(5):
                y = F.conv2d(xd, wd, stride=stride)
                y.backward(g)
            e1.record(); torch.cuda.synchronize()
            tfb = e0.elapsed_time(e1) / 5
            flops = 2 * B * (y.shape[2] * y.shape[3]) * 128 * C * k * k
            print("B%d C%d %dx%d k%d s%d allow_tf32=%s  fwd %.3f ms (%.1f TF/s)  fwd+bwd %.3f ms" %
                  (B, C, H, W, k, stride, tf32, tf, flops / tf / 1e9, tfb))
