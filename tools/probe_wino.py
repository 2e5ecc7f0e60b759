#!/usr/bin/env python3
"""Launch the Winograd convolution kernels of the bench's L2 / L3 layers alone a few times (for rocprofv3 / counters)."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from global_flow_local_attention_amd import _lib, fc_mfma
DEV = "cuda:0"
p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
only_k = int(os.environ.get("PROBE_K", "0"))
for kv in sys.argv[3:]:
    key, val = kv.split("=")
    _lib.set_tuning(int(key), int(val))
    print("tuning", key, val)
for (B, C, H, W, k) in [(32, 128, 64, 44, 5), (32, 256, 32, 22, 3)]:
    if only_k and k != only_k:
        continue
    torch.manual_seed(0)
    s, t = torch.randn(B, C, H, W, device=DEV), torch.randn(B, C, H, W, device=DEV)
    f = torch.randn(B, 2, H, W, device=DEV)
    w0 = torch.randn(128, 2 * C, k, k, device=DEV) * 0.02
    w1 = torch.randn(k * k, 128, device=DEV) * 0.1
    gl = torch.randn(B, k * k, H, W, device=DEV) * 1e-3
    ws = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 0), dtype=torch.uint8, device=DEV)
    sc = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 1), dtype=torch.uint8, device=DEV)
    logits = torch.empty(B, k * k, H, W, device=DEV)
    gs, gt, gf, gw0 = torch.empty_like(s), torch.empty_like(t), torch.empty_like(f), torch.empty_like(w0)
    _lib.call("gfla_fc_forward_f32", s, p(s), p(t), p(f), p(w0), None, p(w1), None, p(ws), p(logits), B, C, H, W, k, 0.1, mode)
    _lib.call("gfla_fc_backward_f32", s, p(ws), p(f), p(w1), p(gl), p(sc), p(gs), p(gt), p(gf), p(gw0), None, None, None, B, C, H, W, k, 0.1, mode, 0)
    for which in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.call("gfla_fc_kernel_f32", s, which, p(ws), p(sc), B, C, H, W, k, mode)
        e0.record()
        for _ in range(iters):
            _lib.call("gfla_fc_kernel_f32", s, which, p(ws), p(sc), B, C, H, W, k, mode)
        e1.record()
        torch.cuda.synchronize()
        print("k%d which %d: %.1f us" % (k, which, e0.elapsed_time(e1) * 1e3 / iters))
