#!/usr/bin/env python3
"""BASELINE config 3 (configs[2]): ExtractorAttn forward only (eval, torch.no_grad) at the attention-layer shapes of a
256x256 image (L3 (256, 32x32) k=3, L2 (128, 64x64) k=5), global batch 32.

    python tools/bench_inference.py [--batch 32] [--iters 10]                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_inference.py --gpus N                                           # batch sharded over N ranks

Legs, each timed on its own (HIP events on the launch stream, MAX over ranks):
  * fused forward: FC layers on this library's MFMA kernels (--fc-mode, default 0 = exact f32; every line carries
    the mode it was measured in) + coefficient-table aggregation, eager and
    captured in a hipGraph; next to the reference's op-by-op composition on the same gfx950 ops;
  * N > 1: all_gather_tiles of the generated feature tiles (the north star's "RCCL all-gather of generated tiles over
    xGMI"), timed separately from the compute and reported as bytes gathered per second.
GFLA_DIST_BACKEND=gloo GFLA_DEVICE=0 lets two ranks share one GPU to exercise the control flow."""
import argparse, json, os, sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import global_flow_local_attention_amd as gfla  # noqa: E402
from global_flow_local_attention_amd import dist as gdist, fc_mfma  # noqa: E402


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


def max_over_ranks(us, world, device):
    if world == 1:
        return us
    t = torch.tensor([us], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return t.item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32, help="GLOBAL batch (sharded over the ranks)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--fc-mode", type=int, choices=(0, 2, 3, 4), default=4,
                    help="arithmetic of the FC contraction: 4 float32 Winograd-domain (the product default), 0 float32 "
                         "direct; 3 / 2 = f16-split operands (labelled experiments)")
    a = ap.parse_args()
    local = int(os.environ.get("GFLA_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    rank, world, _ = gdist.init_from_env(os.environ.get("GFLA_DIST_BACKEND"), device=local)
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    dev = torch.device("cuda", local)
    lo, hi = gdist.shard_range(a.batch, rank, world)
    B = hi - lo
    total, rows = {}, []
    for name, C, H, W, k in (("L3", 256, 32, 32, 3), ("L2", 128, 64, 64, 5)):
        torch.manual_seed(0)  # same parameters on every rank
        m = gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True).to(dev).eval()
        m.fc_mode = a.fc_mode
        g = torch.Generator(device=dev).manual_seed(1 + rank)
        src = torch.randn(B, C, H, W, device=dev, generator=g)
        tgt = torch.randn(B, C, H, W, device=dev, generator=g)
        flow = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(
            torch.randn(B, 2, H, W, device=dev, generator=g) * 12, (3, 3, 3, 3), mode="replicate"), 7, 1).contiguous()
        row = {"layer": name, "fc_mode": a.fc_mode, "B_global": a.batch, "B_rank": B, "C": C, "HxW": "%dx%d" % (H, W), "k": k}
        with torch.no_grad():
            m.fused = True
            out = m(src, tgt, flow)
            row["fused_us"] = round(max_over_ranks(timed(lambda: m(src, tgt, flow), a.iters), world, dev), 1)
            graphed = gfla.graphed_inference(m, (src, tgt, flow))
            row["fused_hipgraph_us"] = round(max_over_ranks(timed(lambda: graphed(src, tgt, flow), a.iters), world, dev), 1)
            if world == 1:
                m.fused = False
                row["op_by_op_us"] = round(timed(lambda: m(src, tgt, flow), max(2, a.iters // 3)), 1)
                m.fused = True
                row["speedup_vs_op_by_op"] = round(row["op_by_op_us"] / row["fused_us"], 2)
            else:
                gathered = gdist.all_gather_tiles(out)
                assert gathered.shape[0] == a.batch
                us = max_over_ranks(timed(lambda: gdist.all_gather_tiles(out), a.iters), world, dev)
                nbytes = gathered.numel() * gathered.element_size()
                row["all_gather_tiles_us"] = round(us, 1)
                row["all_gather_GBps_received_per_rank"] = round(nbytes * (world - 1) / world / us / 1e3, 1)
        for key in ("fused_us", "fused_hipgraph_us", "op_by_op_us", "all_gather_tiles_us"):
            if key in row:
                total[key] = total.get(key, 0) + row[key]
        rows.append(row)
        if rank == 0:
            print(json.dumps(row), flush=True)
    if rank == 0:
        summary = {"fc_mode": a.fc_mode,
                   "fc_arithmetic": fc_mfma.MODE_NAMES[a.fc_mode],
                   "both layers": {k: round(v, 1) for k, v in total.items()}, "n_gpus": world,
                   "images_per_s_fused": round(a.batch / (total["fused_us"] * 1e-6), 1),
                   "images_per_s_fused_hipgraph": round(a.batch / (total["fused_hipgraph_us"] * 1e-6), 1)}
        if "op_by_op_us" in total:
            summary["images_per_s_op_by_op"] = round(a.batch / (total["op_by_op_us"] * 1e-6), 1)
        if "all_gather_tiles_us" in total:
            summary["images_per_s_incl_gather"] = round(a.batch / ((total["fused_us"] + total["all_gather_tiles_us"]) * 1e-6), 1)
        print(json.dumps(summary), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
