#!/usr/bin/env python3
"""Which operand layout makes the FC GEMMs of ExtractorAttn fastest in hipBLASLt (fp32)?"""
import torch
dev = "cuda:0"
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for (name, K, N) in (("L2", 3200, 32 * 64 * 44), ("L3", 2304, 32 * 32 * 22)):
    M = 128
    W = torch.randn(M, K, device=dev); U = torch.randn(K, N, device=dev)
    dH = torch.randn(M, N, device=dev); dHc = torch.randn(N, M, device=dev)
    Wt = W.t().contiguous()
    fl = 2.0 * M * K * N
    res = {}
    res["fwd  H(M,N)  = W @ U"] = t(lambda: torch.mm(W, U))
    res["fwd  Hc(N,M) = U^T @ W^T"] = t(lambda: torch.mm(U.t(), W.t()))
    res["fwd  Hc(N,M) = U^T @ Wt(contig K,M)"] = t(lambda: torch.mm(U.t(), Wt))
    res["dgrad dU(K,N) = W^T @ dH"] = t(lambda: torch.mm(W.t(), dH))
    res["dgrad dU(K,N) = W^T @ dHc^T"] = t(lambda: torch.mm(W.t(), dHc.t()))
    res["dgrad dU(K,N) = Wt @ dH"] = t(lambda: torch.mm(Wt, dH))
    res["wgrad dW(M,K) = dH @ U^T"] = t(lambda: torch.mm(dH, U.t()))
    res["wgrad dW(M,K) = dHc^T @ U^T"] = t(lambda: torch.mm(dHc.t(), U.t()))
    res["wgrad dWt(K,M) = U @ dHc"] = t(lambda: torch.mm(U, dHc))
    res["wgrad dWt(K,M) = U @ dH^T"] = t(lambda: torch.mm(U, dH.t()))
    for k, us in res.items():
        print("%s %-40s %8.1f us  %6.1f TF/s" % (name, k, us, fl / us / 1e6), flush=True)
