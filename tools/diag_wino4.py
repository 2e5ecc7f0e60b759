#!/usr/bin/env python3
"""Order dependence: mode 4 as the FIRST run of a shape in a fresh process, with the allocator's free blocks poisoned."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import global_flow_local_attention_amd as gfla
DEV = "cuda:0"
first = int(sys.argv[1]) if len(sys.argv) > 1 else 4
poison = len(sys.argv) > 2
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
    if poison:
        junk = [torch.full((n,), float("nan"), device=DEV) for n in (300_000_000, 100_000_000, 50_000_000, 20_000_000, 5_000_000, 1_000_000)]
        del junk
    m.fc_mode = mode
    a = [x.to(DEV).requires_grad_() for x in (s, t, f)]
    m.zero_grad()
    out = m(*a)
    out.backward(up.to(DEV))
    torch.cuda.synchronize()
    return [out.detach()] + [x.grad for x in a] + [p.grad.clone() for p in m.parameters()]
names = ["out", "source", "target", "flow", "w0", "b0", "w1", "b1"]
def rel(a, b):
    if not torch.isfinite(a).all(): return float("nan")
    return ((a - b).abs().max() / b.abs().max()).item()
ra = run(first); rb = run(4 - first); rc = run(first)
print("first=%d poison=%d" % (first, poison))
print("  first vs other :", " ".join("%s %.2e" % (n, rel(a, b)) for n, a, b in zip(names, ra, rb)))
print("  first vs again :", " ".join("%s %.2e" % (n, rel(a, b)) for n, a, b in zip(names, ra, rc)))
