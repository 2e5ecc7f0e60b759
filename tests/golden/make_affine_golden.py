#!/usr/bin/env python3
"""Golden vectors for AffineRegularizationLoss, produced by the REFERENCE's own class
(model/networks/external_function.py:31-77): its __init__ / __call__ / calculate_loss / flow2grid run unchanged on the
host, with its two custom ops replaced by the CPU oracle (oracle/cpu_modules.py -- the literal restatement of the
reference kernels, itself pinned to the real kernels by ref_golden.npz).  Values and d/d flow for kz = 3, 5 incl. a
non-square field.  Needs /root/reference; run in the build container:  python tests/golden/make_affine_golden.py
"""
import os, sys, types
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import global_flow_local_attention_amd as gfla  # noqa: E402
from oracle import cpu_modules, cpu_oracle  # noqa: E402

sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
sys.modules.setdefault("torchvision.models", types.ModuleType("torchvision.models"))
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
gfla.install("/root/reference", fuse_extractor_attn=False)
import model.networks.external_function as ef  # noqa: E402

CASES = [("kz3_12x10", 3, 2, 12, 10, 1.5), ("kz5_16x11", 5, 2, 16, 11, 2.5), ("kz3_7x9_b1", 3, 1, 7, 9, 6.0)]


class _Extractor(torch.nn.Module):  # block_extractor.py:45-54 on the host
    def __init__(self, k):
        super().__init__()
        self.k = k

    def forward(self, source, flow_field):
        return cpu_modules._BlockExtractorCPU.apply(source, flow_field, self.k)


class _Reshape(torch.nn.Module):  # local_attn_reshape.py:40-46 on the host
    def forward(self, x, k):
        return cpu_modules._LocalAttnReshapeCPU.apply(x, k)


def main():
    cpu_oracle.build()
    out = {}
    for i, (name, kz, B, H, W, std) in enumerate(CASES):
        g = torch.Generator().manual_seed(300 + i)
        flow = (torch.randn(B, 2, H, W, generator=g) * std).requires_grad_()
        ref = ef.AffineRegularizationLoss(kz)                       # the reference's constructor (projector kernel)
        ref.extractor, ref.reshape = _Extractor(kz), _Reshape()
        loss = ref(flow)                                            # the reference's __call__ / calculate_loss
        loss.backward()
        out[name + "/flow"] = flow.detach().numpy()
        out[name + "/loss"] = loss.detach().numpy()
        out[name + "/g_flow"] = flow.grad.numpy()
        print(name, float(loss))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "affine_golden.npz"), **out)


if __name__ == "__main__":
    main()
