#!/usr/bin/env python3
"""Timing ablations of fc_wino16_wgrad_kernel (`make PROBES=1` builds only; tuning key 20 = 64 + bits; results are garbage)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import global_flow_local_attention_amd as gfla
import bench
dev = torch.device("cuda", 0)
hp = bench.HotPath(32, dev, seed=100, fc_impl="mfma", fc_mode=5)
hp.two_streams = False
rs = gfla.Resample2d(4, 1, 2)
hp.step(rs, allreduce=False)
names = {0: "everything", 1: "no transform", 2: "no MFMAs / A reads", 3: "neither", 4: "no lift / split", 6: "no split, no MFMA", 7: "loop + dY loads + staging only", 8: "no dY loads", 15: "skeleton"}
for key in (0, 1, 2, 3, 4, 6, 7, 8, 15, 0):
    gfla.set_tuning(20, 64 + key if key else 0)
    rows = bench.fc_kernel_probes(hp)
    r = [x for x in rows if x["dims"][-1] == 5 and "weight-grad source + target" in x["kernel"]]
    print(json.dumps({"key": key, "what": names[key], "us": r[0]["avg_us"]}), flush=True)
gfla.set_tuning(20, 0)
