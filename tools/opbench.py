#!/usr/bin/env python3
"""Op-level micro-benchmark: every C-ABI entry point at the BASELINE shapes, HIP-event timed,
next to the REAL reference kernels (oracle/_ref, when built) on the same GPU and inputs.

    python tools/opbench.py [--iters 20] [--out gpurun_out/opbench.jsonl] [--only be_fwd,agg_fwd] [--tuning key=val,...]

Prints one JSON object per case: algorithmic MB, avg us, GB/s, fraction of the 8 TB/s HBM peak,
and the reference kernel's time for the same call.
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import global_flow_local_attention_amd as gfla  # noqa: E402
from global_flow_local_attention_amd import _lib  # noqa: E402
from bench import algorithmic_bytes, HBM_PEAK_GBS  # noqa: E402

DEV = "cuda:0"


def flow_of(kind, B, H, W):
    g = torch.Generator(device=DEV).manual_seed(7)
    if kind == "zero":
        return torch.zeros(B, 2, H, W, device=DEV)
    n = torch.randn(B, 2, H, W, device=DEV, generator=g)
    if kind == "smooth":
        return F.avg_pool2d(F.pad(n * 12, (3, 3, 3, 3), mode="replicate"), 7, 1).contiguous()
    if kind == "coherent":
        return n * 2
    return n * 8  # wild


def time_fn(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default="")
    ap.add_argument("--tuning", default="")
    ap.add_argument("--no-ref", action="store_true")
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)
    for kv in (x for x in args.tuning.split(",") if x):
        k, v = kv.split("=")
        gfla.set_tuning(int(k), int(v))
    ref = None
    if not args.no_ref:
        from oracle import ref_ext
        ref = ref_ext if ref_ext.available() else None
    L = _lib.lib()
    rows = []

    def emit(case, name, plain_args, us, ref_us=None, note=""):
        nbytes = algorithmic_bytes(name, plain_args)
        gbs = nbytes / (us * 1e-6) / 1e9
        row = {"case": case, "entry": name, "alg_MB": round(nbytes / 1e6, 2), "us": round(us, 1),
               "GBps": round(gbs, 1), "frac_peak": round(gbs / HBM_PEAK_GBS, 4),
               "ref_us": None if ref_us is None else round(ref_us, 1),
               "speedup_vs_ref": None if ref_us is None else round(ref_us / us, 2), "tuning": args.tuning, "note": note}
        rows.append(row)
        print(json.dumps(row), flush=True)

    def want(tag):
        return not only or tag in only

    # ---- HBM copy yardstick ---------------------------------------------------------------
    if want("copy"):
        a = torch.empty(256 * 1024 * 1024, device=DEV)  # 1 GiB
        b = torch.empty_like(a)
        us = time_fn(lambda: b.copy_(a), args.iters)
        gbs = 2 * a.numel() * 4 / (us * 1e-6) / 1e9
        print(json.dumps({"case": "torch copy 1GiB (read+write)", "us": round(us, 1), "GBps": round(gbs, 1),
                          "frac_peak": round(gbs / HBM_PEAK_GBS, 4)}), flush=True)
        us = time_fn(lambda: b.zero_(), args.iters)
        gbs = a.numel() * 4 / (us * 1e-6) / 1e9
        print(json.dumps({"case": "torch fill 1GiB (write only)", "us": round(us, 1), "GBps": round(gbs, 1),
                          "frac_peak": round(gbs / HBM_PEAK_GBS, 4)}), flush=True)
        del a, b

    be_cases = [("cfg2", 1, 64, 256, 176, 3), ("cfg2", 1, 64, 256, 176, 5),
                ("cfg3-L3", 32, 256, 32, 32, 3), ("cfg3-L2", 32, 128, 64, 64, 5),
                ("b32x176-L3", 32, 256, 32, 22, 3), ("b32x176-L2", 32, 128, 64, 44, 5)]
    for (tag, B, C, H, W, k) in be_cases:
        src = torch.randn(B, C, H, W, device=DEV)
        for kind in ("zero", "smooth", "wild"):
            flow = flow_of(kind, B, H, W)
            case = "%s B%d C%d %dx%d k%d flow=%s" % (tag, B, C, H, W, k, kind)
            out = torch.empty(B, C, k * H, k * W, device=DEV)
            pa = ("ptr", "ptr", "ptr", B, C, H, W, H, W, k)
            if want("be_fwd"):
                fn = lambda: _lib.call("gfla_block_extractor_fwd_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(out), B, C, H, W, H, W, k)
                us = time_fn(fn, args.iters)
                rus = None
                if ref and kind != "zero":
                    m = ref._mod("block_extractor_cuda")
                    rus = time_fn(lambda: m.forward(src, flow, out, k), max(3, args.iters // 4))
                emit("be_fwd " + case, "gfla_block_extractor_fwd_f32", pa, us, rus)
            if want("be_bwd") and kind != "wild":
                gout = torch.randn(B, C, k * H, k * W, device=DEV)
                gs, gf = torch.zeros_like(src), torch.zeros_like(flow)
                pb = ("ptr", "ptr", "ptr", "ptr", "ptr", B, C, H, W, H, W, k)
                fn = lambda: _lib.call("gfla_block_extractor_bwd_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(gout), _lib.ptr(gs), _lib.ptr(gf), B, C, H, W, H, W, k)
                us = time_fn(fn, max(3, args.iters // 2))
                rus = None
                if ref and kind == "smooth":
                    m = ref._mod("block_extractor_cuda")
                    rus = time_fn(lambda: m.backward(src, flow, gout, gs, gf, k), 3)
                emit("be_bwd " + case, "gfla_block_extractor_bwd_f32", pb, us, rus)
                if kind == "zero":  # the ExtractorAttn target call: no flow gradient wanted
                    pc = ("ptr", "ptr", "ptr", "ptr", None, B, C, H, W, H, W, k)
                    fn = lambda: _lib.call("gfla_block_extractor_bwd_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(gout), _lib.ptr(gs), None, B, C, H, W, H, W, k)
                    emit("be_bwd(src only) " + case, "gfla_block_extractor_bwd_f32", pc, time_fn(fn, max(3, args.iters // 2)))
                del gout, gs, gf
            if want("be_unfold") and tag != "cfg2" and kind == "smooth":
                outu = torch.empty(B, C * k * k, H, W, device=DEV)
                fn = lambda: _lib.call("gfla_block_extractor_unfold_fwd_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(outu), B, C, H, W, H, W, k, 1)
                emit("be_unfold_fwd " + case, "gfla_block_extractor_fwd_f32", pa, time_fn(fn, args.iters))
                gs, gf = torch.zeros_like(src), torch.zeros_like(flow)
                fn = lambda: _lib.call("gfla_block_extractor_unfold_bwd_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(outu), _lib.ptr(gs), _lib.ptr(gf), B, C, H, W, H, W, k, 1)
                emit("be_unfold_bwd " + case, "gfla_block_extractor_bwd_f32", ("ptr",) * 5 + (B, C, H, W, H, W, k), time_fn(fn, max(3, args.iters // 2)))
                # the GEMM that consumes it (W: 128 x C*k*k)
                wmat = torch.randn(128, C * k * k, device=DEV)
                us = time_fn(lambda: torch.mm(wmat, outu.view(C * k * k, B * H * W)), max(3, args.iters // 2))
                fl = 2.0 * B * H * W * 128 * C * k * k
                print(json.dumps({"case": "fc gemm on unfold " + case, "us": round(us, 1), "TFLOPs": round(fl / us / 1e6, 1)}), flush=True)
                del outu, gs, gf, wmat
            del out
        del src
        torch.cuda.empty_cache()

    for (tag, B, C, H, W, k) in [c for c in be_cases if c[0] != "cfg2"]:
        src = torch.randn(B, C, H, W, device=DEV)
        lg = torch.randn(B, k * k, H, W, device=DEV)
        out, attn = torch.empty_like(src), torch.empty_like(lg)
        for kind in ("zero", "smooth", "wild"):
            flow = flow_of(kind, B, H, W)
            case = "%s B%d C%d %dx%d k%d flow=%s" % (tag, B, C, H, W, k, kind)
            # the entry points the PRODUCT runs: coefficient-table forward / matrix-core scatter backward, both with
            # caller-owned scratch (the plain entry points select round 1's kernels)
            fws = torch.empty(max(int(L.gfla_aggregate_fwd_workspace_bytes(B, H, W, k)), 16), dtype=torch.uint8, device=DEV)
            bws = torch.empty(max(int(L.gfla_scatter_workspace_bytes(B, H, W, (k + 1) ** 2)), 16), dtype=torch.uint8, device=DEV)
            if want("agg_fwd"):
                pa = ("ptr",) * 5 + (B, C, H, W, H, W, k, 1)
                fn = lambda: _lib.call("gfla_local_attn_aggregate_fwd_ws_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(lg), _lib.ptr(out), _lib.ptr(attn), _lib.ptr(fws), B, C, H, W, H, W, k, 1)
                emit("agg_fwd " + case, "gfla_local_attn_aggregate_fwd_f32", pa, time_fn(fn, args.iters), note="_ws entry (what the product calls)")
            if want("agg_bwd") and kind == "smooth":
                _lib.call("gfla_local_attn_aggregate_fwd_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(lg), _lib.ptr(out), _lib.ptr(attn), B, C, H, W, H, W, k, 1)
                go = torch.randn_like(src)
                gs, gf, gl = torch.zeros_like(src), torch.zeros_like(flow), torch.zeros_like(lg)
                pb = ("ptr",) * 7 + (B, C, H, W, H, W, k, 1)
                fn = lambda: _lib.call("gfla_local_attn_aggregate_bwd_ws_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(attn), _lib.ptr(go), _lib.ptr(gs), _lib.ptr(gf), _lib.ptr(gl), _lib.ptr(bws), B, C, H, W, H, W, k, 1)
                emit("agg_bwd " + case, "gfla_local_attn_aggregate_bwd_f32", pb, time_fn(fn, max(3, args.iters // 2)), note="_ws entry (what the product calls)")
        if want("reshape"):
            o2 = torch.empty(B, 1, k * H, k * W, device=DEV)
            fn = lambda: _lib.call("gfla_local_attn_reshape_fwd_f32", lg, _lib.ptr(lg), _lib.ptr(o2), B, H, W, k)
            rus = None
            if ref:
                m = ref._mod("local_attn_reshape_cuda")
                rus = time_fn(lambda: m.forward(lg, o2, k), 5)
            emit("reshape_fwd %s" % tag, "gfla_local_attn_reshape_fwd_f32", ("ptr", "ptr", B, H, W, k), time_fn(fn, args.iters), rus)
        del src, lg, out, attn
        torch.cuda.empty_cache()

    rs_cases = [("cfg2", 1, 64, 256, 176), ("vgg relu3_1 256^2", 32, 256, 64, 64), ("vgg relu4_1 256^2", 32, 512, 32, 32),
                ("vgg relu3_1 x176", 32, 256, 64, 44), ("vgg relu4_1 x176", 32, 512, 32, 22)]
    for (tag, B, C, H, W) in rs_cases:
        i1 = torch.randn(B, C, H, W, device=DEV)
        for kind in ("smooth", "wild"):
            i2 = torch.cat((flow_of(kind, B, H, W), torch.full((B, 1, H, W), 2.0, device=DEV)), 1).contiguous()
            case = "%s B%d C%d %dx%d k4 flow=%s" % (tag, B, C, H, W, kind)
            out = torch.empty_like(i1)
            if want("rs_fwd"):
                fn = lambda: _lib.call("gfla_resample2d_fwd_f32", i1, _lib.ptr(i1), _lib.ptr(i2), _lib.ptr(out), B, C, H, W, H, W, 4, 1)
                rus = None
                if ref:
                    m = ref._mod("resample2d_cuda")
                    rus = time_fn(lambda: m.forward(i1, i2, out, 4, 1), 5)
                emit("rs_fwd " + case, "gfla_resample2d_fwd_f32", ("ptr",) * 3 + (B, C, H, W, H, W, 4, 1), time_fn(fn, args.iters), rus)
            if want("rs_bwd") and kind == "smooth":
                go = torch.randn_like(i1)
                g1, g2 = torch.zeros_like(i1), torch.zeros_like(i2)
                fn = lambda: _lib.call("gfla_resample2d_bwd_f32", i1, _lib.ptr(i1), _lib.ptr(i2), _lib.ptr(go), _lib.ptr(g1), _lib.ptr(g2), B, C, H, W, H, W, 4, 1, 1)
                rus = None
                if ref:
                    m = ref._mod("resample2d_cuda")
                    rus = time_fn(lambda: m.backward(i1, i2, go, g1, g2, 4, 1), 3)
                emit("rs_bwd " + case, "gfla_resample2d_bwd_f32", ("ptr",) * 5 + (B, C, H, W, H, W, 4, 1, 1), time_fn(fn, max(3, args.iters // 2)), rus)
                fn1 = lambda: _lib.call("gfla_resample2d_bwd_f32", i1, _lib.ptr(i1), _lib.ptr(i2), _lib.ptr(go), _lib.ptr(g1), None, B, C, H, W, H, W, 4, 1, 1)
                emit("rs_bwd(in1 only) " + case, "gfla_resample2d_bwd_f32", ("ptr",) * 4 + (None, B, C, H, W, H, W, 4, 1, 1), time_fn(fn1, max(3, args.iters // 2)))
                fn2 = lambda: _lib.call("gfla_resample2d_bwd_f32", i1, _lib.ptr(i1), _lib.ptr(i2), _lib.ptr(go), None, _lib.ptr(g2), B, C, H, W, H, W, 4, 1, 1)
                emit("rs_bwd(in2 only) " + case, "gfla_resample2d_bwd_f32", ("ptr",) * 3 + (None, "ptr", B, C, H, W, H, W, 4, 1, 1), time_fn(fn2, max(3, args.iters // 2)))
        torch.cuda.empty_cache()

    if want("graph"):
        import time
        for (C, H, W, k) in ((256, 32, 22, 3), (128, 64, 44, 5)):
            m = gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True).to(DEV).eval()
            inp = (torch.randn(1, C, H, W, device=DEV), torch.randn(1, C, H, W, device=DEV), flow_of("smooth", 1, H, W))
            with torch.no_grad():
                eager = time_fn(lambda: m(*inp), 50)
            g = gfla.graphed_inference(m, inp)
            graph = time_fn(lambda: g.graph.replay(), 50)
            print(json.dumps({"case": "ExtractorAttn B1 C%d %dx%d k%d inference" % (C, H, W, k), "eager_us": round(eager, 1),
                              "hipgraph_us": round(graph, 1), "speedup": round(eager / graph, 2)}), flush=True)

    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "a") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
