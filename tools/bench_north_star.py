#!/usr/bin/env python3
"""Sweep of the two forward ops the north star prices against the HBM roofline (bench.py: op_roofline) over the
library's tuning keys -- one JSON line per configuration.

    python tools/bench_north_star.py [--iters 20] [--sweep be|agg|abl|none] [--face] [--flow smooth|zero]

--sweep be : block_extractor forward variants (key 0 kernel, key 4 planes per workgroup, key 24 threads, key 25
             non-temporal stores, key 5 split), each checked against round 1's kernel (max abs difference)
"""
import argparse
import itertools
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


def be_check(layers, B):
    """max |new - round-1 kernel| per layer (same inputs)."""
    res = {}
    gen = torch.Generator(device=DEV).manual_seed(5)
    for (name, C, H, W, k) in layers:
        src = torch.randn(B, C, H, W, device=DEV, generator=gen)
        flow = bench.smooth_flow(B, H, W, DEV, gen)
        outs = []
        for key0 in (2, 0):
            gfla.set_tuning(0, key0)
            o = torch.empty(B, C, k * H, k * W, device=DEV)
            _lib.call("gfla_block_extractor_fwd_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(o), B, C, H, W, H, W, k)
            outs.append(o)
        gfla.set_tuning(0, 0)
        res[name] = float((outs[0] - outs[1]).abs().max())
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sweep", default="be")
    ap.add_argument("--face", action="store_true")
    ap.add_argument("--flow", default="smooth")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    layers = bench.FACE_LAYERS if args.face else bench.LAYERS
    B = args.batch

    def run(tag, keys):
        for kk, v in keys.items():
            gfla.set_tuning(kk, v)
        r = bench.op_roofline(DEV, B=B, iters=args.iters, layers=layers, flow_kind=args.flow)
        for kk in keys:
            gfla.set_tuning(kk, 0)
        line = {"tag": tag, "keys": {str(a): b for a, b in keys.items()}}
        for name, d in r["layers"].items():
            line[name] = {"be_us": d["block_extractor_fwd"]["us"], "be_frac": d["block_extractor_fwd"]["frac"],
                          "agg_us": d["local_attn_fwd"]["us"], "agg_frac": d["local_attn_fwd"]["frac"],
                          "pair_frac": d["pair"]["frac"]}
        print(json.dumps(line), flush=True)

    print(json.dumps({"check_max_abs_vs_round1_kernel": be_check(layers, B)}), flush=True)
    run("round-1 kernel (lane = output quad)", {0: 2})
    run("default", {})
    if args.sweep == "be":
        run("pix direct stores (key 0 = 3)", {0: 3})
        run("wave-per-flow-row (key 0 = 4)", {0: 4})
        for G in (1, 2, 3, 4, 8):
            for thr in (256, 384, 512, 704, 1024):
                run("wrow G=%d threads=%d" % (G, thr), {0: 4, 4: G, 24: thr})
        for kb in (52, 64, 100, 150):
            run("wrow LDS budget %d KB" % kb, {0: 4, 10: kb})
            run("wrow LDS budget %d KB threads=1024" % kb, {0: 4, 10: kb, 24: 1024})
    if args.sweep == "agg":   # softmax + aggregate forward: workgroup size, planes per chunk, channel ranges, tile width
        for thr in (768, 704, 512, 384, 256):
            for ch in (0, 2):
                for ns in (0, 1, 2, 4, 8):
                    run("agg threads=%d CH=%d ranges=%d" % (thr, ch, ns), {9: thr, 4: ch, 5: ns})
        for tw in (8, 16, 32):
            run("agg tile width %d" % tw, {16: tw})
            run("agg tile width %d threads=384" % tw, {16: tw, 9: 384})
        run("agg k=3 on the record/stream path (key 8 = 2)", {8: 2})
        for thr in (384, 512, 704):
            run("agg k=3 stream path threads=%d" % thr, {8: 2, 9: thr})
    if args.sweep == "abl":   # needs a `make PROBES=1` library
        for abl in (0, 1, 2, 3):
            run("wave-per-flow-row, ablation %d (1 = no patch reads, 2 = no stores)" % abl, {0: 4, 27: abl})
        for abl in (0, 2):
            run("pix direct, ablation %d" % abl, {0: 3, 27: abl})


if __name__ == "__main__":
    main()
