#!/usr/bin/env python3
"""Why is windowed be_bwd slow with smooth flows?  Times cfg2 backward for a ladder of flows."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from global_flow_local_attention_amd import _lib
import global_flow_local_attention_amd as gfla
from opbench import flow_of, time_fn
DEV = "cuda:0"
B, C, H, W = 1, 64, 256, 176
src = torch.randn(B, C, H, W, device=DEV)
for k in (3, 5):
    gout = torch.randn(B, C, k * H, k * W, device=DEV)
    base = flow_of("smooth", B, H, W)
    flows = {"zero": torch.zeros_like(base), "const0.5": torch.full_like(base, 0.5), "const0.5+int3": torch.full_like(base, 3.5),
             "smooth*0.01": base * 0.01, "smooth*0.3": base * 0.3, "smooth*1": base, "smooth*1 x-only": torch.cat((base[:, :1], torch.zeros_like(base[:, :1])), 1).contiguous(),
             "smooth*1 y-only": torch.cat((torch.zeros_like(base[:, :1]), base[:, 1:]), 1).contiguous()}
    for name, fl in flows.items():
        fl = fl.contiguous()
        for need_flow in (True, False):
            gs, gf = torch.zeros_like(src), torch.zeros_like(fl)
            fn = lambda: _lib.call("gfla_block_extractor_bwd_f32", src, _lib.ptr(src), _lib.ptr(fl), _lib.ptr(gout), _lib.ptr(gs), _lib.ptr(gf) if need_flow else None, B, C, H, W, H, W, k)
            print("k=%d %-18s gflow=%d  %8.1f us   max|flow|=%.2f" % (k, name, need_flow, time_fn(fn, 5), fl.abs().max().item()), flush=True)
