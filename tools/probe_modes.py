#!/usr/bin/env python3
"""Per-kernel times of the FC path in several arithmetic modes (bench.fc_kernel_probes)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import global_flow_local_attention_amd as gfla
import bench
dev = torch.device("cuda", 0)
for mode in [int(m) for m in sys.argv[1:]] or [5, 2]:
    hp = bench.HotPath(32, dev, seed=100, fc_impl="mfma", fc_mode=mode)
    hp.two_streams = False
    rs = gfla.Resample2d(4, 1, 2)
    hp.step(rs, allreduce=False)
    for x in bench.fc_kernel_probes(hp):
        print(mode, x["dims"][-1], x["kernel"].split(": ")[1][:40], x["avg_us"], flush=True)
