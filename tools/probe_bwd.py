#!/usr/bin/env python3
"""The FC backward of both bench layers a few times (for rocprofv3 --kernel-trace): usage probe_bwd.py [KEY VALUE]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import global_flow_local_attention_amd as gfla
from global_flow_local_attention_amd import _lib, fc_mfma
import bench
if len(sys.argv) > 2:
    gfla.set_tuning(int(sys.argv[1]), int(sys.argv[2]))
dev = torch.device("cuda", 0)
hp = bench.HotPath(32, dev, seed=100, fc_impl="mfma", fc_mode=5)
for mod, (src, tgt, flow) in zip(hp.attn, hp.inputs):
    B, C, H, W = src.shape
    k = mod.kernel_size
    fc = mod.fully_connect_layer
    with torch.no_grad():
        s, t, f = src.detach().contiguous(), tgt.detach().contiguous(), flow.detach().contiguous()
        w0, w1 = fc[0].weight.detach().contiguous(), fc[2].weight.detach().reshape(k * k, 128).contiguous()
        ws = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, 5, 0), dtype=torch.uint8, device=dev)
        sc = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, 5, 1), dtype=torch.uint8, device=dev)
        lg = s.new_empty(B, k * k, H, W)
        gl = torch.randn(lg.shape, device=dev) * 1e-3
        gs, gt, gf = torch.empty_like(s), torch.empty_like(t), torch.empty_like(f)
        gw0 = torch.empty_like(w0)
        for _ in range(5):
            _lib.call("gfla_fc_forward_f32", s, _lib.ptr(s), _lib.ptr(t), _lib.ptr(f), _lib.ptr(w0), _lib.ptr(fc[0].bias),
                      _lib.ptr(w1), _lib.ptr(fc[2].bias), _lib.ptr(ws), _lib.ptr(lg), B, C, H, W, k, 0.1, 5)
            _lib.call("gfla_fc_backward_f32", s, _lib.ptr(ws), _lib.ptr(f), _lib.ptr(w1), _lib.ptr(gl), _lib.ptr(sc),
                      _lib.ptr(gs), _lib.ptr(gt), _lib.ptr(gf), _lib.ptr(gw0), None, None, None, B, C, H, W, k, 0.1, 5, 0)
        torch.cuda.synchronize()
