#!/usr/bin/env python3
"""gfla_fc_backward_f32 in mode 4 vs mode 0 on identical inputs, with and without the accumulate flags."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from global_flow_local_attention_amd import _lib, fc_mfma
DEV = "cuda:0"
p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
B, C, H, W, k = 32, 128, 64, 44, 5
torch.manual_seed(0)
s, t = torch.randn(B, C, H, W, device=DEV), torch.randn(B, C, H, W, device=DEV)
f = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(torch.randn(B, 2, H, W, device=DEV) * 12, (3, 3, 3, 3), mode="replicate"), 7, 1).contiguous()
w0 = torch.randn(128, 2 * C, k, k, device=DEV) / (2 * C * k * k) ** 0.5
b0 = torch.where(torch.arange(128, device=DEV) % 2 == 0, 8.0, -8.0) + torch.randn(128, device=DEV) * 0.1
w1 = torch.randn(k * k, 128, device=DEV) / 128 ** 0.5
b1 = torch.randn(k * k, device=DEV) * 0.1
gl = torch.randn(B, k * k, H, W, device=DEV) * 1e-3
res = {}
for mode in (0, 4):
    for flags in (0, 3):
        ws = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 0), dtype=torch.uint8, device=DEV)
        logits = torch.empty(B, k * k, H, W, device=DEV)
        _lib.call("gfla_fc_forward_f32", s, p(s), p(t), p(f), p(w0), p(b0), p(w1), p(b1), p(ws), p(logits), B, C, H, W, k, 0.1, mode)
        sc = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 1), dtype=torch.uint8, device=DEV)
        base_s = torch.full((B, C, H, W), 0.5, device=DEV) if flags else torch.empty(B, C, H, W, device=DEV)
        base_f = torch.full((B, 2, H, W), 0.25, device=DEV) if flags else torch.empty(B, 2, H, W, device=DEV)
        gs, gf = base_s.clone(), base_f.clone()
        gt = torch.empty(B, C, H, W, device=DEV)
        gw0, gb0, gw1, gb1 = torch.empty_like(w0), torch.empty_like(b0), torch.empty_like(w1), torch.empty_like(b1)
        _lib.call("gfla_fc_backward_f32", f, p(ws), p(f), p(w1), p(gl), p(sc), p(gs), p(gt), p(gf), p(gw0), p(gb0), p(gw1), p(gb1),
                  B, C, H, W, k, 0.1, mode, flags)
        torch.cuda.synchronize()
        if flags:
            gs, gf = gs - 0.5, gf - 0.25
        res[(mode, flags)] = (logits, gs, gt, gf, gw0, gb0, gw1, gb1)
names = ("logits", "g_source", "g_target", "g_flow", "g_w0", "g_b0", "g_w1", "g_b1")
rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
for flags in (0, 3):
    print("flags %d:" % flags, " ".join("%s %.2e" % (n, rel(a, b)) for n, a, b in zip(names, res[(4, flags)], res[(0, flags)])))
print("mode0 flags3 vs flags0:", " ".join("%s %.2e" % (n, rel(a, b)) for n, a, b in zip(names, res[(0, 3)], res[(0, 0)])))
print("mode4 flags3 vs flags0:", " ".join("%s %.2e" % (n, rel(a, b)) for n, a, b in zip(names, res[(4, 3)], res[(4, 0)])))
