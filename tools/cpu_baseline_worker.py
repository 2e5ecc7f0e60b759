#!/usr/bin/env python3
"""One worker of bench.py's cpu_baseline: the reference composition with the oracle's kernels on `--threads` host threads pinned
to the CPUs `--cpus a-b`, batch 1, for `--budget` seconds; prints {"steps": n, "seconds": dt}.  (The oracle is the checker and the
CPU yardstick only -- bench.py's cpu_baseline leg is one of the three places allowed to run it.)"""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--cpus", default="")
ap.add_argument("--budget", type=float, default=15.0)
ap.add_argument("--start-at", type=float, default=0.0, help="time.time() at which every worker starts its timed loop")
a = ap.parse_args()
if a.cpus:
    lo, hi = (int(x) for x in a.cpus.split("-"))
    try:
        os.sched_setaffinity(0, set(range(lo, hi + 1)))
    except (AttributeError, OSError):
        pass
os.environ["OMP_NUM_THREADS"] = str(a.threads)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.set_num_threads(a.threads)
import bench  # noqa: E402
from oracle import cpu_modules, cpu_oracle  # noqa: E402

cpu_oracle.build()
cpu_oracle.set_threads(a.threads)
mods = [cpu_modules.ExtractorAttnCPU(C, k, torch.nn.LeakyReLU(0.1), softmax=True) for (_, C, _, _, k) in bench.LAYERS]
hp = bench.HotPath(1, "cpu", seed=0, modules=mods)
res = cpu_modules.Resample2dCPU(4, 1, 2)
hp.step(res, allreduce=False)  # warm-up
while time.time() < a.start_at:
    time.sleep(0.01)
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < a.budget:
    hp.step(res, allreduce=False)
    n += 1
print(json.dumps({"steps": n, "seconds": time.perf_counter() - t0}), flush=True)
