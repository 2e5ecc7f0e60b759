#!/usr/bin/env python3
"""Mode 4 (Winograd) against mode 0 (direct), one half of the layer at a time: where do they differ?"""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from global_flow_local_attention_amd import _lib, fc_mfma
DEV = "cuda:0"
p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())

def run(mode, B, C, H, W, k, is_source, x, w0, dG):
    g = fc_mfma.geometry(H, W, k, is_source)
    ws = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 0), dtype=torch.uint8, device=DEV)
    out = torch.zeros((B, g["Mg"], 128), device=DEV)
    _lib.call("gfla_fc_conv_fwd_f32", x, p(x), p(w0), is_source, p(ws), p(out), B, C, H, W, k, mode)
    rows = (torch.arange(g["Ho"])[:, None] * g["Wp"] + torch.arange(g["Wo"])[None, :]).reshape(-1).to(DEV)
    z = torch.zeros(B, g["Sz"], 128, device=DEV)
    z[:, g["lead"] + rows, :] = dG.permute(0, 2, 3, 1).reshape(B, -1, 128)
    sc = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 1), dtype=torch.uint8, device=DEV)
    gx = torch.zeros((B, C, H, W), device=DEV)
    gw = torch.zeros((128, 2 * C, k, k), device=DEV)
    _lib.call("gfla_fc_conv_bwd_f32", x, p(z), is_source, p(ws), p(sc), p(gx), p(gw), B, C, H, W, k, mode)
    torch.cuda.synchronize()
    return out[:, :g["Ho"] * g["Wo"]].reshape(B, g["Ho"], g["Wo"], 128), gx, gw, g

for (B, C, H, W, k) in [(2, 128, 64, 44, 5), (33, 128, 64, 44, 5), (2, 128, 64, 64, 5), (2, 256, 32, 22, 3)]:
    for is_source in (1, 0):
        torch.manual_seed(0)
        x = torch.randn(B, C, H, W, device=DEV)
        w0 = torch.randn(128, 2 * C, k, k, device=DEV) * 0.02
        g = fc_mfma.geometry(H, W, k, is_source)
        dG = torch.randn(B, 128, g["Ho"], g["Wo"], device=DEV) * 1e-3
        o0, gx0, gw0, _ = run(0, B, C, H, W, k, is_source, x, w0, dG)
        o4, gx4, gw4, _ = run(4, B, C, H, W, k, is_source, x, w0, dG)
        rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
        print("B%d C%d %dx%d k%d source=%d: fwd %.2e  grad_x %.2e  grad_w %.2e" % (B, C, H, W, k, is_source, rel(o4, o0), rel(gx4, gx0), rel(gw4, gw0)))
        d = (gx4 - gx0).abs()
        if rel(gx4, gx0) > 1e-4:
            thr = 1e-3 * gx0.abs().max()
            bad = d > thr
            print("   bad elements %d of %d; per-b %s" % (int(bad.sum()), bad.numel(), bad.flatten(1).sum(1).tolist()[:8]))
            print("   bad rows y:", bad.any(3).any(1).any(0).nonzero().flatten().tolist())
            print("   bad cols x:", bad.any(2).any(1).any(0).nonzero().flatten().tolist())
            print("   bad channels (first 16):", bad.any(3).any(2).any(0).nonzero().flatten().tolist()[:16])
