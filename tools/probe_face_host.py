#!/usr/bin/env python3
"""Is the bf16 face leg bound by the host (launch rate) or by the GPU?  Host time to ENQUEUE one step (step() returns, no
synchronisation) next to the time until the GPU has finished it, for the two-stream and the one-stream evaluation."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for dual in (True, False):
    fp = bench.FacePath(8, dev, seed=3, frames=6, dual_stream=dual)
    for _ in range(3):
        fp.step(allreduce=False)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fp.step(allreduce=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3)
        tot.append((t2 - t0) * 1e3)
    print(json.dumps({"dual_stream": dual, "host_enqueue_ms": round(min(enq), 2), "until_gpu_done_ms": round(min(tot), 2)}), flush=True)
