#!/usr/bin/env python3
"""Time gfla_max_cosine_fwd_f32 against the reference's formulation (normalise -> bmm -> max, torch on the
same GPU) at the sampling-correctness-loss shapes.  usage: python tools/bench_max_cosine.py [--iters N]"""
import argparse, json, os, sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import global_flow_local_attention_amd as gfla  # noqa: E402

SHAPES = [  # name, B, C, N (VGG features of a 256x176 / 256x256 image)
    ("relu4_1 32x22", 32, 512, 32 * 22),
    ("relu3_1 64x44", 32, 256, 64 * 44),
    ("relu4_1 32x32", 32, 512, 32 * 32),
    ("relu3_1 64x64", 32, 256, 64 * 64),
    ("relu2_1 128x88", 8, 128, 128 * 88),
]


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


def reference_form(s, t, eps=1e-8):
    sa = s.transpose(1, 2)
    sn = sa / (sa.norm(dim=2, keepdim=True) + eps)
    tn = t / (t.norm(dim=1, keepdim=True) + eps)
    return torch.max(torch.bmm(sn, tn), dim=1)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--only", default="", help="substring of the shape name")
    ap.add_argument("--loss", action="store_true", help="also time PerceptualCorrectness.calculate_loss")
    ap.add_argument("--tuning", default="", help="key=value,... passed to gfla.set_tuning")
    a = ap.parse_args()
    for kv in filter(None, a.tuning.split(",")):
        k, v = kv.split("=")
        gfla.set_tuning(int(k), int(v))
    for name, B, C, N in SHAPES:
        if a.only not in name:
            continue
        g = torch.Generator(device="cuda").manual_seed(0)
        s = torch.randn(B, C, N, device="cuda", generator=g).relu_()
        t = torch.randn(B, C, N, device="cuda", generator=g).relu_()
        us = timed(lambda: gfla.max_cosine_similarity(s, t), a.iters)
        flops = 2.0 * B * C * N * N
        row = {"shape": name, "B": B, "C": C, "N": N, "us": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1),
               "frac_f32_mfma_peak": round(flops / us / 1e6 / 157.3, 3)}
        if not a.no_ref:
            try:
                ref_us = timed(lambda: reference_form(s, t), max(2, a.iters // 3))
                row.update(torch_bmm_max_us=round(ref_us, 1), speedup=round(ref_us / us, 2),
                           max_abs_diff=float((gfla.max_cosine_similarity(s, t) - reference_form(s, t)).abs().max()))
            except RuntimeError as e:  # the [B,N,N] matrix may not fit
                row["torch_bmm_max_us"] = "failed: %s" % str(e)[:60]
        print(json.dumps(row), flush=True)
        if a.loss:   # the whole calculate_loss (resample -> map -> mean), fused map vs torch ops, fwd + bwd
            H, W = {704: (32, 22), 2816: (64, 44), 1024: (32, 32), 4096: (64, 64), 11264: (128, 88)}[N]
            flow = (torch.randn(B, 2, H, W, device="cuda", generator=g) * 2).requires_grad_()
            mod = gfla.PerceptualCorrectness()
            mod.target_vgg, mod.source_vgg = {"f": t.view(B, C, H, W)}, {"f": s.view(B, C, H, W)}
            out = {"shape": name, "what": "calculate_loss fwd+bwd (flow gradient only)"}
            for fused in (True, False):
                mod.fused = fused

                def step():
                    flow.grad = None
                    mod.calculate_loss(flow, "f").backward()
                out["fused_us" if fused else "torch_ops_us"] = round(timed(step, a.iters), 1)
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
