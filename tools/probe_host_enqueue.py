#!/usr/bin/env python3
"""Host enqueue time vs GPU time of the headline step: K steps issued back to back, clock read before the final synchronize
(host done) and after it (GPU done).  If the two are close the step is host-bound at its call boundaries."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import global_flow_local_attention_amd as gfla
import bench

dev = torch.device("cuda", 0)
try:
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
except AttributeError:
    pass
for two in (True, False):
    hp = bench.HotPath(32, dev, seed=100, fc_impl="mfma", fc_mode=5)
    hp.two_streams = two
    rs = gfla.Resample2d(4, 1, 2)
    for _ in range(5):
        hp.step(rs, allreduce=False)
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        hp.step(rs, allreduce=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"two_streams": two, "host_enqueue_ms_per_step": round((t1 - t0) / K * 1e3, 3),
                      "gpu_done_ms_per_step": round((t2 - t0) / K * 1e3, 3)}), flush=True)
