#!/usr/bin/env python3
"""Timing ablations of fc_wino16_conv_kernel<5> (`make PROBES=1` builds only; tuning key 20 compiles parts out, results are
garbage): the forward of both halves at the bench shape, HIP events around 10 launches."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import global_flow_local_attention_amd as gfla
from global_flow_local_attention_amd import _lib, fc_mfma
import bench

dev = torch.device("cuda", 0)
hp = bench.HotPath(32, dev, seed=100, fc_impl="mfma", fc_mode=5)
hp.two_streams = False
rs = gfla.Resample2d(4, 1, 2)
hp.step(rs, allreduce=False)
names = {0: "everything", 1: "no transform", 2: "no MFMAs / A reads", 3: "neither", 4: "no B reloads", 8: "no staging",
         32: "no f16 split", 64: "no epilogue", 67: "no epilogue, transform, MFMA", 128: "epilogue without global stores", 256: "epilogue without exchange / barriers", 384: "epilogue: partials only", 79: "loop skeleton only (67 + no B, no staging)"}
for key in (0, 1, 2, 3, 64, 67, 256, 384):
    gfla.set_tuning(20, key)
    rows = bench.fc_kernel_probes(hp)
    r = [x for x in rows if x["dims"][-1] == 5 and "one launch" in x["kernel"]]
    print(json.dumps({"key": key, "what": names[key], "fwd_both_us": r[0]["avg_us"], "dgrad_both_us": r[1]["avg_us"]}), flush=True)
gfla.set_tuning(20, 0)
