#!/usr/bin/env python3
"""ExtractorAttn module, mode 4 vs mode 0, bench shape: which gradients differ, and does the scatter path matter?"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import global_flow_local_attention_amd as gfla
DEV = "cuda:0"
B, C, H, W, k = 32, 128, 64, 44, 5
torch.manual_seed(0)
s, t = torch.randn(B, C, H, W), torch.randn(B, C, H, W)
f = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(torch.randn(B, 2, H, W) * 12, (3, 3, 3, 3), mode="replicate"), 7, 1).contiguous()
up = torch.randn(B, C, H, W)
m = gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True)
with torch.no_grad():
    m.fully_connect_layer[0].bias.copy_(torch.where(torch.arange(128) % 2 == 0, 8.0, -8.0))
m = m.to(DEV)
def run(mode):
    m.fc_mode = mode
    a = [x.to(DEV).requires_grad_() for x in (s, t, f)]
    m.zero_grad()
    out = m(*a)
    out.backward(up.to(DEV))
    torch.cuda.synchronize()
    return [out.detach()] + [x.grad for x in a] + [p.grad.clone() for p in m.parameters()]
names = ["out", "source", "target", "flow", "w0", "b0", "w1", "b1"]
rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
for key14 in (0, 1):
    gfla.set_tuning(14, key14)
    r0 = run(0); r4 = run(4); r0b = run(0); r4b = run(4)
    print("tuning14=%d  mode4 vs mode0:" % key14, " ".join("%s %.2e" % (n, rel(a, b)) for n, a, b in zip(names, r4, r0)))
    print("             mode0 vs mode0:", " ".join("%s %.2e" % (n, rel(a, b)) for n, a, b in zip(names, r0b, r0)))
    print("             mode4 vs mode4:", " ".join("%s %.2e" % (n, rel(a, b)) for n, a, b in zip(names, r4b, r4)))
