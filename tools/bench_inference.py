#!/usr/bin/env python3
"""BASELINE config 3: ExtractorAttn forward only (eval, torch.no_grad), B=32, 256x256 image shapes
(L3 (256, 32x32) k=3, L2 (128, 64x64) k=5), fused evaluation vs the reference's op-by-op composition on the
same gfx950 ops, eager and captured in a hipGraph.  usage: python tools/bench_inference.py [--batch 32]"""
import argparse, json, os, sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import global_flow_local_attention_amd as gfla  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    gfla.enable_gemm_tuning("/tmp/gfla_tunableop_inference.csv")
    B = a.batch
    total = {}
    for name, C, H, W, k in (("L3", 256, 32, 32, 3), ("L2", 128, 64, 64, 5)):
        torch.manual_seed(0)
        m = gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True).cuda().eval()
        g = torch.Generator(device="cuda").manual_seed(1)
        src = torch.randn(B, C, H, W, device="cuda", generator=g)
        tgt = torch.randn(B, C, H, W, device="cuda", generator=g)
        flow = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(
            torch.randn(B, 2, H, W, device="cuda", generator=g) * 12, (3, 3, 3, 3), mode="replicate"), 7, 1).contiguous()
        row = {"layer": name, "B": B, "C": C, "HxW": "%dx%d" % (H, W), "k": k}
        with torch.no_grad():
            m.fused = True
            m(src, tgt, flow)  # tuning / lazy init
            row["fused_us"] = round(timed(lambda: m(src, tgt, flow), a.iters), 1)
            graphed = gfla.graphed_inference(m, (src, tgt, flow))
            row["fused_hipgraph_us"] = round(timed(lambda: graphed(src, tgt, flow), a.iters), 1)
            m.fused = False
            row["op_by_op_us"] = round(timed(lambda: m(src, tgt, flow), max(2, a.iters // 3)), 1)
            m.fused = True
        row["speedup_vs_op_by_op"] = round(row["op_by_op_us"] / row["fused_us"], 2)
        for key in ("fused_us", "fused_hipgraph_us", "op_by_op_us"):
            total[key] = total.get(key, 0) + row[key]
        print(json.dumps(row), flush=True)
    print(json.dumps({"both layers": {k: round(v, 1) for k, v in total.items()},
                      "images_per_s_fused": round(B / (total["fused_us"] * 1e-6), 1),
                      "images_per_s_op_by_op": round(B / (total["op_by_op_us"] * 1e-6), 1)}), flush=True)


if __name__ == "__main__":
    main()
