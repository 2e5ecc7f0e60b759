#!/usr/bin/env python3
"""Does PyTorch's TunableOp find faster GEMM solutions for the FC shapes of the bench?  Times the three
GEMMs of the L2 source half (fwd, grad operand, grad weight) and the L3 ones with and without tuning and
writes the chosen solutions to gpurun_out/tunableop_results.csv."""
import os, sys, time
import torch

def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters

SHAPES = [("L2", 128, 3200, 32 * 64 * 44), ("L3", 128, 2304, 32 * 32 * 22)]

def run(tag):
    out = {}
    for name, M, K, N in SHAPES:
        W = torch.randn(M, K, device="cuda"); U = torch.randn(K, N, device="cuda"); G = torch.randn(M, N, device="cuda")
        out[name + " fwd  W@U"] = timed(lambda: torch.mm(W, U))
        out[name + " gU   W^T@G"] = timed(lambda: torch.mm(W.t(), G))
        out[name + " gW   G@U^T"] = timed(lambda: torch.mm(G, U.t()))
    for k, v in out.items():
        M, K, N = next((m, k_, n) for nm, m, k_, n in SHAPES if k.startswith(nm))
        print("%-8s %-14s %8.1f us  %6.1f TF/s" % (tag, k, v, 2.0 * M * K * N / v / 1e6), flush=True)
    return out

base = run("default")
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "200")))
torch.cuda.tunable.set_max_tuning_iterations(20)
os.makedirs("gpurun_out", exist_ok=True)
torch.cuda.tunable.set_filename("gpurun_out/tunableop_results.csv")
t0 = time.time()
tuned = run("tuning")
print("tuning pass took %.1f s" % (time.time() - t0))
tuned = run("tuned")
torch.cuda.tunable.write_file()
for r in torch.cuda.tunable.get_results():
    print(r)
