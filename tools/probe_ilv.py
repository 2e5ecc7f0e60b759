#!/usr/bin/env python3
"""A/B of a tuning key on the FC kernels: time per internal kernel (bench.fc_kernel_probes) and the largest difference of the
layer's logits between the two settings.  usage: probe_ilv.py KEY VALUE [VALUE ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import global_flow_local_attention_amd as gfla
from global_flow_local_attention_amd import _lib, fc_mfma
import bench

key = int(sys.argv[1])
vals = [0] + [int(v) for v in sys.argv[2:]]
dev = torch.device("cuda", 0)
hp = bench.HotPath(32, dev, seed=100, fc_impl="mfma", fc_mode=5)
hp.two_streams = False
rs = gfla.Resample2d(4, 1, 2)
hp.step(rs, allreduce=False)


calls = {}


def logits():
    outs = []
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
            gl = torch.randn(lg.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 1e-3
            gs, gt, gf = torch.empty_like(s), torch.empty_like(t), torch.empty_like(f)
            gw0 = torch.empty_like(w0)
            _lib.call("gfla_fc_forward_f32", s, _lib.ptr(s), _lib.ptr(t), _lib.ptr(f), _lib.ptr(w0), _lib.ptr(fc[0].bias),
                      _lib.ptr(w1), _lib.ptr(fc[2].bias), _lib.ptr(ws), _lib.ptr(lg), B, C, H, W, k, 0.1, 5)
            _lib.call("gfla_fc_backward_f32", s, _lib.ptr(ws), _lib.ptr(f), _lib.ptr(w1), _lib.ptr(gl), _lib.ptr(sc),
                      _lib.ptr(gs), _lib.ptr(gt), _lib.ptr(gf), _lib.ptr(gw0), None, None, None, B, C, H, W, k, 0.1, 5, 0)
            torch.cuda.synchronize()
            outs += [lg.clone(), gs.clone(), gt.clone(), gw0.clone(), gf.clone()]
            st = torch.cuda.current_stream(dev)
            for name, fn in (("fwd", lambda: _lib.call("gfla_fc_forward_f32", s, _lib.ptr(s), _lib.ptr(t), _lib.ptr(f), _lib.ptr(w0), _lib.ptr(fc[0].bias),
                                                      _lib.ptr(w1), _lib.ptr(fc[2].bias), _lib.ptr(ws), _lib.ptr(lg), B, C, H, W, k, 0.1, 5)),
                             ("bwd", lambda: _lib.call("gfla_fc_backward_f32", s, _lib.ptr(ws), _lib.ptr(f), _lib.ptr(w1), _lib.ptr(gl), _lib.ptr(sc),
                                                      _lib.ptr(gs), _lib.ptr(gt), _lib.ptr(gf), _lib.ptr(gw0), None, None, None, B, C, H, W, k, 0.1, 5, 0))):
                fn(); fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(10):
                    fn()
                e1.record(st)
                torch.cuda.synchronize()
                calls["%s k%d" % (name, k)] = round(e0.elapsed_time(e1) * 100, 1)
    return outs


base = None
for v in vals:
    gfla.set_tuning(key, v)
    o = logits()
    if base is None:
        base = o
    diff = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(o, base)]
    rows = bench.fc_kernel_probes(hp)
    r = {x["kernel"].split(": ")[1][:28] + " k%d" % x["dims"][-1]: x["avg_us"] for x in rows if "one launch" in x["kernel"]}
    print(json.dumps({"key": key, "value": v, "rel_diff_vs_0": ["%.1e" % d for d in diff], "us": r, "call_us": dict(calls)}), flush=True)
gfla.set_tuning(key, 0)
