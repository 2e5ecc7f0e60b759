#!/usr/bin/env python3
"""block_extractor forward (reference layout) at (32, 128, 64, W), k = 5 over map widths W: does the write stream depend
on how output rows (5 W floats) line up with 64- / 128-byte segments?  One JSON line per (W, kernel variant)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import global_flow_local_attention_amd as gfla  # noqa: E402
from global_flow_local_attention_amd import _lib  # noqa: E402

DEV = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, C, H, k = 32, 128, 64, 5
VARIANTS = [("quad (round 1)", {0: 2}), ("wrow", {0: 4}), ("pix direct", {0: 3}), ("auto", {})]
stream = torch.cuda.current_stream(DEV)
for W in (32, 40, 44, 48, 52, 56, 60, 64, 72, 80, 96):
    gen = torch.Generator(device=DEV).manual_seed(3)
    src = torch.randn(B, C, H, W, device=DEV, generator=gen)
    flow = bench.smooth_flow(B, H, W, DEV, gen)
    out = torch.empty(B, C, k * H, k * W, device=DEV)
    nbytes = 4 * (B * C * H * W + 2 * B * H * W + B * C * k * k * H * W)
    row = {"W": W, "row_bytes": 4 * k * W, "alg_MB": round(nbytes / 1e6, 1)}
    for name, keys in VARIANTS:
        for kk, v in keys.items():
            gfla.set_tuning(kk, v)
        fn = lambda: _lib.call("gfla_block_extractor_fwd_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(out), B, C, H, W, H, W, k)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        for kk in keys:
            gfla.set_tuning(kk, 0)
        row[name] = {"us": round(us, 1), "TBps": round(nbytes / us / 1e6, 2)}
    print(json.dumps(row), flush=True)
