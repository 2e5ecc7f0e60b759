#!/usr/bin/env python3
"""Matrix-core scatter paths (csrc/patch_mfma.hip) vs the LDS-atomic kernels at the bench shapes, HIP-event timed.

    python tools/bench_scatter.py [--iters 20] [--rows 0,1,2,3,4] [--flows smooth,coherent,wild]
Band height R = tuning key 13 (0 = the library's choice)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from global_flow_local_attention_amd import _lib  # noqa: E402
from opbench import flow_of, time_fn  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rows", default="0,1,2,4")
    ap.add_argument("--flows", default="zero,smooth,coherent,wild")
    ap.add_argument("--tuning", default="", help="key=value,... applied before anything runs")
    args = ap.parse_args()
    for kv in filter(None, args.tuning.split(",")):
        _lib.lib().gfla_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
    lib = _lib.lib()
    B = 32
    for kind in args.flows.split(","):
        for (name, C, H, W, k) in (("attn2", 128, 64, 44, 5), ("attn3", 256, 32, 22, 3)):
            src = torch.randn(B, C, H, W, device=DEV)
            flow = flow_of(kind, B, H, W)
            attn = torch.softmax(torch.randn(B, k * k, H, W, device=DEV), 1)
            go = torch.randn(B, C, H, W, device=DEV)
            gs = torch.zeros_like(src)
            ws = _lib.scatter_workspace(src, B, H, W, (k + 1) ** 2)
            tail = (B, C, H, W, H, W, k, 1)

            def run(w):
                _lib.call("gfla_local_attn_aggregate_bwd_ws_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(attn),
                          _lib.ptr(go), _lib.ptr(gs), None, None, _lib.ptr(w), *tail)
            row = {"op": "aggregate d/d source", "shape": name, "flow": kind, "lds_atomic_us": round(time_fn(lambda: run(None), args.iters), 1)}
            row["adaptive_us"] = round(time_fn(lambda: run(ws), args.iters), 1)
            torch.cuda.synchronize()
            lib.gfla_set_tuning(15, 100000)  # always the matrix-core path
            for r in args.rows.split(","):
                lib.gfla_set_tuning(13, int(r))
                row["mfma_R%s_us" % r] = round(time_fn(lambda: run(ws), args.iters), 1)
            lib.gfla_set_tuning(13, 0)
            lib.gfla_set_tuning(15, 0)
            print(json.dumps(row), flush=True)
        for (name, C, H, W) in (("relu3_1", 256, 64, 44), ("relu4_1", 512, 32, 22)):
            i1 = torch.randn(B, C, H, W, device=DEV)
            flow = flow_of(kind, B, H, W)
            i2 = torch.cat((flow, torch.full((B, 1, H, W), 2.0, device=DEV)), 1).contiguous()
            go = torch.randn(B, C, H, W, device=DEV)
            g1 = torch.zeros_like(i1)
            ws = _lib.scatter_workspace(i1, B, H, W, 16)
            tail = (B, C, H, W, H, W, 4, 1, 1)

            def run(w):
                _lib.call("gfla_resample2d_bwd_ws_f32", i1, _lib.ptr(i1), _lib.ptr(i2), _lib.ptr(go), _lib.ptr(g1), None,
                          _lib.ptr(w), *tail)
            g_fix = None
            run(None)
            g_fix = g1.clone()
            row = {"op": "resample2d d/d input1", "shape": name, "flow": kind, "lds_atomic_us": round(time_fn(lambda: run(None), args.iters), 1)}
            lib.gfla_set_tuning(23, 1)      # round 1's double planes instead of the fixed-point ones
            g1.zero_()
            run(None)
            row["fix_vs_f64_maxdiff"] = float((g1 - g_fix).abs().max() / g1.abs().max())
            row["lds_atomic_f64_us"] = round(time_fn(lambda: run(None), args.iters), 1)
            lib.gfla_set_tuning(23, 0)
            g1.zero_()
            run(ws)
            row["ws_vs_nows_maxdiff"] = float((g1 - g_fix).abs().max() / g_fix.abs().max())
            row["adaptive_us"] = round(time_fn(lambda: run(ws), args.iters), 1)
            torch.cuda.synchronize()
            lib.gfla_set_tuning(15, 100000)
            for r in args.rows.split(","):
                lib.gfla_set_tuning(13, int(r))
                row["mfma_R%s_us" % r] = round(time_fn(lambda: run(ws), args.iters), 1)
            lib.gfla_set_tuning(13, 0)
            lib.gfla_set_tuning(15, 0)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
