#!/usr/bin/env python3
"""BASELINE configs[1] op by op (bench.config2_ops) with optional tuning keys, one JSON line per row.

    python tools/bench_config2.py [--tuning 30=1,...] [--tag name] [--flows smooth,wild] [--no-ref] [--out file.jsonl]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import global_flow_local_attention_amd as gfla  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tuning", default="")
    ap.add_argument("--tag", default="")
    ap.add_argument("--flows", default="smooth,zero,wild,integer,near_integer,oob")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--split", action="store_true", help="also time each gradient of the backward calls alone")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    for kv in filter(None, a.tuning.split(",")):
        gfla.set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
    res = bench.config2_ops(torch.device("cuda", 0), iters=a.iters, flows=tuple(a.flows.split(",")), with_ref=not a.no_ref, split=a.split)
    lines = [json.dumps(dict(r, tag=a.tag, tuning=a.tuning)) for r in res["rows"]]
    for ln in lines:
        print(ln, flush=True)
    print(json.dumps({"tag": a.tag, "slower_than_reference_on": res["slower_than_reference_on"]}), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "a") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
