#!/usr/bin/env python3
"""Local-attention forward (softmax + aggregate) at the attention-layer shapes: the paired-read kernel
(agg_fwd_pk_kernel, default) next to round 1's ds_read_b32 kernel (tuning key 8 = 1), HIP-event timed, with the
fraction of the HBM roofline (algorithmic bytes / time / 8 TB/s) and the max abs difference between the two.

    python tools/bench_agg_fwd.py [--iters 20] [--flows smooth,coherent,wild] [--g 0,4,8] [--face]
--g: channels per workgroup to try (tuning key 4; 0 = the library's choice)."""
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
    ap.add_argument("--flows", default="smooth,coherent,wild")
    ap.add_argument("--g", default="0")
    ap.add_argument("--pitch", type=int, default=0, help="tuning key 17: LDS row-pair pitch in words (experiment)")
    ap.add_argument("--wide-tail", type=int, default=0)
    ap.add_argument("--dbg", type=int, default=0)
    ap.add_argument("--ns", type=int, default=0, help="tuning key 5: channel ranges per sample")
    ap.add_argument("--tile", type=int, default=0, help="tuning key 16: tile width 8 / 16 / 32 (0 = library choice)")
    ap.add_argument("--threads", type=int, default=0, help="tuning key 9: workgroup size cap of the paired-read kernel")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--face", action="store_true", help="256x256 image shapes instead of 256x176")
    ap.add_argument("--dtype", default="f32", choices=("f32", "bf16"))
    args = ap.parse_args()
    lib = _lib.lib()
    lib.gfla_set_tuning(9, args.threads)
    lib.gfla_set_tuning(17, args.pitch)
    lib.gfla_set_tuning(16, args.tile)
    lib.gfla_set_tuning(5, args.ns)
    lib.gfla_set_tuning(18, args.dbg)
    B = args.batch
    shapes = (("attn2", 128, 64, 64 if args.face else 44, 5), ("attn3", 256, 32, 32 if args.face else 22, 3))
    dt = torch.float32 if args.dtype == "f32" else torch.bfloat16
    esz = 4 if args.dtype == "f32" else 2
    for kind in args.flows.split(","):
        for (name, C, H, W, k) in shapes:
            src = torch.randn(B, C, H, W, device=DEV).to(dt)
            flow = flow_of(kind, B, H, W).to(dt)
            logits = torch.randn(B, k * k, H, W, device=DEV).to(dt)
            out, attn = torch.empty_like(src), torch.empty_like(logits)
            nbytes = esz * (2 * B * C * H * W + 2 * B * H * W + 2 * B * k * k * H * W)

            def run():
                _lib.aggregate_fwd(src, flow, logits, out, attn, k, True)
            row = {"op": "local-attn forward", "shape": name, "dims": [B, C, H, W, k], "flow": kind, "alg_MB": round(nbytes / 1e6, 1)}
            lib.gfla_set_tuning(8, 1)
            us = time_fn(run, args.iters)
            ref_out, ref_attn = out.clone(), attn.clone()
            row["b32_kernel_us"] = round(us, 1)
            row["b32_frac_hbm"] = round(nbytes / us / 1e6 / 8, 3)
            lib.gfla_set_tuning(8, 0)
            for g in args.g.split(","):
                lib.gfla_set_tuning(4, int(g))
                us = time_fn(run, args.iters)
                tag = "pk_G%s" % g if int(g) else "pk"
                row[tag + "_us"] = round(us, 1)
                row[tag + "_frac_hbm"] = round(nbytes / us / 1e6 / 8, 3)
                row[tag + "_maxdiff"] = float((out.float() - ref_out.float()).abs().max())
                row[tag + "_attn_equal"] = bool(torch.equal(attn, ref_attn))
            lib.gfla_set_tuning(4, 0)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
