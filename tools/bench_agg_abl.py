#!/usr/bin/env python3
"""Timing ablations of the softmax + aggregate forward (agg_coef_kernel + agg_fwd_stream_kernel): the default library and
the variants of tools/ubench/build_agg_abl.sh, one process each (GFLA_HIP_LIBRARY), HIP-event timed at the north star's L2
shape.  Variant results are garbage by construction; only the times mean anything."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "everything", 1: "no LDS reads", 2: "no prefetch / staging", 4: "no barrier", 6: "no staging, no barrier",
         8: "no arithmetic", 9: "no reads, no arithmetic", 15: "chunk loop empty (prologue + stores)"}

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    from global_flow_local_attention_amd import _lib
    from opbench import flow_of, time_fn
    B, C, H, W, k = 32, 128, 64, 44, 5
    src = torch.randn(B, C, H, W, device="cuda:0")
    flow = flow_of("smooth", B, H, W)
    logits = torch.randn(B, k * k, H, W, device="cuda:0")
    out, attn = torch.empty_like(src), torch.empty_like(logits)
    run = lambda: _lib.aggregate_fwd(src, flow, logits, out, attn, k, True)
    us = min(time_fn(run, 30) for _ in range(3))
    print(json.dumps({"us": round(us, 1)}))
    sys.exit(0)

libs = [(0, None)] + sorted((int(os.path.basename(f)[len("libgfla_agg_abl"):-3]), f)
                            for f in glob.glob(os.path.join(ROOT, "tools", "ubench", "abl", "libgfla_agg_abl*.so")))
for bits, path in libs:
    env = dict(os.environ)
    if path:
        env["GFLA_HIP_LIBRARY"] = path
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True)
    try:
        us = json.loads(r.stdout.strip().splitlines()[-1])["us"]
    except Exception:
        us = None
        sys.stderr.write(r.stderr[-500:])
    print(json.dumps({"abl": bits, "what": NAMES.get(bits, "?"), "local_attn_fwd_L2_us": us}), flush=True)
