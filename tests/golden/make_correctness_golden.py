#!/usr/bin/env python3
"""Golden vectors for the sampling-correctness loss, produced by the REFERENCE's own
`PerceptualCorrectness.calculate_loss` (model/networks/external_function.py:246-279) executed on the
host: the method is called unbound on a stand-in `self` whose `resample` is the CPU restatement of
Resample2d (oracle/cpu_modules.py -- itself pinned to the real reference kernels by ref_golden.npz).
Needs /root/reference; run in the build container:  python tests/golden/make_correctness_golden.py
"""
import os, sys, types
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import global_flow_local_attention_amd as gfla  # noqa: E402
from oracle.cpu_modules import Resample2dCPU  # noqa: E402

sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
sys.modules.setdefault("torchvision.models", types.ModuleType("torchvision.models"))
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
gfla.install("/root/reference", fuse_extractor_attn=False)
import model.networks.external_function as ef  # noqa: E402

CASES = [  # name, B, C, H, W, flow std, masked
    ("c16_12x10", 2, 16, 12, 10, 1.5, False),
    ("c24_9x13_mask", 3, 24, 9, 13, 2.5, True),
    ("c64_16x11", 1, 64, 16, 11, 4.0, False),
]


def inputs(name, B, C, H, W, std, masked, seed):
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(B, C, H, W, generator=g).relu() + 0.05 * torch.rand(B, C, H, W, generator=g)
    tgt = torch.randn(B, C, H, W, generator=g).relu() + 0.05 * torch.rand(B, C, H, W, generator=g)
    flow = torch.randn(B, 2, 2 * H, 2 * W, generator=g) * std      # interpolated down to (H, W) inside
    mask = torch.sigmoid(torch.randn(B, 1, 2 * H, 2 * W, generator=g)) if masked else None
    return src, tgt, flow, mask


def main():
    out = {}
    for i, (name, B, C, H, W, std, masked) in enumerate(CASES):
        src, tgt, flow, mask = inputs(name, B, C, H, W, std, masked, 100 + i)
        src.requires_grad_()
        tgt.requires_grad_()
        flow.requires_grad_()
        me = types.SimpleNamespace(target_vgg={"f": tgt}, source_vgg={"f": src}, eps=1e-8,
                                   resample=Resample2dCPU(4, 1, sigma=2))
        loss = ef.PerceptualCorrectness.calculate_loss(me, flow, "f", mask)
        loss.backward()
        # the reference's bmm/max intermediates, for the op-level check
        s = src.detach().view(B, C, -1).transpose(1, 2)
        t = tgt.detach().view(B, C, -1)
        corr = torch.bmm(s / (s.norm(dim=2, keepdim=True) + 1e-8), t / (t.norm(dim=1, keepdim=True) + 1e-8))
        best, idx = corr.max(dim=1)
        for key, val in (("src", src), ("tgt", tgt), ("flow", flow), ("loss", loss), ("g_src", src.grad),
                         ("g_tgt", tgt.grad), ("g_flow", flow.grad), ("best", best), ("idx", idx)):
            out["%s/%s" % (name, key)] = val.detach().numpy()
        if mask is not None:
            out["%s/mask" % name] = mask.numpy()
        print(name, float(loss))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "correctness_golden.npz"), **out)


if __name__ == "__main__":
    main()
