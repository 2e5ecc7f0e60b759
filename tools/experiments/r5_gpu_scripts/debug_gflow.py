"""Localise the d/dflow mismatch of be_bwd_tile_kernel at (1,64,256,176) on smooth flows (session 1: rel err 0.35)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import global_flow_local_attention_amd as gfla
from global_flow_local_attention_amd import _lib
from oracle import ref_ext
from util import make_flow, randn
DEV = "cuda:0"
B, C, H, W = 1, 64, 256, 176
for k in (3,):
    s, f = randn((B, C, H, W), seed=1210).to(DEV), make_flow("smooth", B, H, W, seed=1211).to(DEV)
    up = randn((B, C, k * H, k * W), seed=1212).to(DEV)
    ws, wf = ref_ext.block_extractor_bwd(s.double(), f.double(), up.double(), k)
    def run(keys, need_src=True, Csub=C):
        olds = {a: gfla.set_tuning(a, b) for a, b in keys.items()}
        gs, gf = torch.zeros_like(s), torch.zeros_like(f)
        _lib.call("gfla_block_extractor_bwd_f32", s, _lib.ptr(s), _lib.ptr(f), _lib.ptr(up), _lib.ptr(gs) if need_src else None,
                  _lib.ptr(gf), B, C, H, W, H, W, k)
        torch.cuda.synchronize()
        for a, b in olds.items():
            gfla.set_tuning(a, b)
        return gs, gf
    for name, keys, ns in (("default", {}, True), ("flow only", {}, False), ("G=1", {34: 1}, True), ("G=64", {34: 64}, True), ("G=2", {34: 2}, True),
                           ("old global atomics", {2: 1, 30: 1}, True), ("round-1 window", {30: 1}, True), ("tiny LDS (rounds)", {10: 16}, True),
                           ("tile 8x64", {31: 8, 32: 64}, True), ("tile 16x16", {31: 16, 32: 16}, True)):
        gs, gf = run(keys, ns)
        e = (gf.double() - wf).abs()
        bad = e > 1e-3 * wf.abs().max()
        ys, xs = bad[0].any(0).nonzero(as_tuple=True)
        print("%-20s max err %.3e (max|want| %.3e)  bad pixels %d  rows [%s..%s] cols [%s..%s]  gsrc err %.2e" % (
            name, e.max().item(), wf.abs().max().item(), int(bad[0].any(0).sum()), ys.min().item() if len(ys) else "-", ys.max().item() if len(ys) else "-",
            xs.min().item() if len(xs) else "-", xs.max().item() if len(xs) else "-", (gs.double() - ws).abs().max().item() if ns else -1))
        if len(ys) and name == "default":
            cnt = bad[0].any(0)
            print("   bad per tile row (16):", [int(cnt[i * 16:(i + 1) * 16].sum()) for i in range(16)])
            print("   bad per column block (30):", [int(cnt[:, i * 30:(i + 1) * 30].sum()) for i in range(6)])
            idx = bad[0].any(0).nonzero()[:12].tolist()
            for (yy, xx) in idx:
                print("   (%d,%d) flow (%.3f, %.3f) got (%.4f, %.4f) want (%.4f, %.4f)" % (yy, xx, f[0, 0, yy, xx].item(), f[0, 1, yy, xx].item(),
                      gf[0, 0, yy, xx].item(), gf[0, 1, yy, xx].item(), wf[0, 0, yy, xx].item(), wf[0, 1, yy, xx].item()))
