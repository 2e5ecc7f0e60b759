#!/usr/bin/env python3
"""Per-wave phase times of the Winograd convolution kernel (DBG = 16 instantiation, s_memtime)."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from global_flow_local_attention_amd import _lib, fc_mfma
DEV = "cuda:0"
p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
B, C, H, W, k, mode = 32, 128, 64, 44, 5, int(sys.argv[1]) if len(sys.argv) > 1 else 4
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
stamps = torch.zeros(4096, 8, 6, dtype=torch.int64, device=DEV)
_lib.lib().gfla_fc_wino_debug_buffer(p(stamps))
_lib.set_tuning(20, 16)
for which in (0,):
    for _ in range(2):
        stamps.zero_()
        _lib.call("gfla_fc_kernel_f32", s, which, p(ws), p(sc), B, C, H, W, k, mode)
        torch.cuda.synchronize()
    st = stamps.cpu().double()
    used = st[:, :, 1].sum(1) > 0
    st = st[used]
    print("which %d: %d workgroups" % (which, st.shape[0]))
    for xh in (0, 1):
        sel = st[:, 4 * xh:4 * xh + 4, :]
        m = sel.mean((0, 1))
        print("  waves xh=%d (first half = %s): prologue %.0f  first %.0f/step  second %.0f/step  barrier wait %.0f/step  epilogue %.0f   total %.0f" %
              (xh, "multiply" if xh == 0 else "transform", m[0], m[1] / 16, m[2] / 16, m[3] / 16, m[4], m[0] + m[1] + m[2] + m[3] + m[4]))
_lib.set_tuning(20, 0)
_lib.lib().gfla_fc_wino_debug_buffer(None)
