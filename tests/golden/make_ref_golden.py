#!/usr/bin/env python3
"""Generates tests/golden/ref_golden.npz from the REAL reference kernels.

Run on a GPU box (the reference extensions in oracle/_ref/ are the reference's own
block_extractor_cuda / local_attn_reshape_cuda / resample2d_cuda, compiled unmodified from
/root/reference by oracle/build_ref.sh):

    gpurun -- 'python tests/golden/make_ref_golden.py gpurun_out/ref_golden.npz'

then copy gpurun_out/ref_golden.npz to tests/golden/.  Inputs are seeded (tests/util.py) and stored
next to the outputs, so the file is self-contained: the CPU suite checks the oracle against it
(tests/test_golden_cpu.py) and the GPU suite checks the gfx950 kernels against it.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from util import make_flow, rand, randn  # noqa: E402
from oracle import ref_ext  # noqa: E402


def main(out_path):
    assert ref_ext.available(), "oracle/_ref is not built"
    dev = "cuda:0"
    z = {}
    # block_extractor: flow reaches out of bounds; Hs != Hf case as used by AffineRegularizationLoss
    s, f = randn((2, 4, 12, 10), seed=101), make_flow("wild", 2, 12, 10, seed=102)
    z["be_source"], z["be_flow"] = s.numpy(), f.numpy()
    for k in (3, 5):
        out = ref_ext.block_extractor_fwd(s.to(dev), f.to(dev), k)
        g = randn(tuple(out.shape), seed=103 + k)
        gs, gf = ref_ext.block_extractor_bwd(s.to(dev), f.to(dev), g.to(dev), k)
        z["be_out_k%d" % k] = out.cpu().numpy()
        z["be_gout_k%d" % k] = g.numpy()
        z["be_gsrc_k%d" % k] = gs.cpu().numpy()
        z["be_gflow_k%d" % k] = gf.cpu().numpy()
    s2, f2 = randn((2, 1, 12, 9), seed=110), torch.zeros(2, 2, 10, 7) + 1.0
    z["be2_source"], z["be2_flow"] = s2.numpy(), f2.numpy()
    z["be2_out_k3"] = ref_ext.block_extractor_fwd(s2.to(dev), f2.to(dev), 3).cpu().numpy()
    # fp64 instantiation
    s3, f3 = randn((1, 2, 7, 6), torch.float64, seed=111), make_flow("coherent", 1, 7, 6, torch.float64, seed=112)
    z["be3_source"], z["be3_flow"] = s3.numpy(), f3.numpy()
    z["be3_out_k3"] = ref_ext.block_extractor_fwd(s3.to(dev), f3.to(dev), 3).cpu().numpy()
    # local_attn_reshape
    x = randn((2, 9, 7, 5), seed=120)
    z["lar_in"] = x.numpy()
    out = ref_ext.local_attn_reshape_fwd(x.to(dev), 3)
    z["lar_out"] = out.cpu().numpy()
    g = randn(tuple(out.shape), seed=121)
    z["lar_gout"] = g.numpy()
    z["lar_gin"] = ref_ext.local_attn_reshape_bwd(x.to(dev), g.to(dev), 3).cpu().numpy()
    # resample2d (k=4, dilation 1 as in PerceptualCorrectness; per-pixel sigma; negative coordinates)
    i1 = randn((2, 5, 9, 8), seed=130)
    i2 = torch.cat((make_flow("wild", 2, 9, 8, seed=131), rand((2, 1, 9, 8), seed=132) * 2 + 0.5), 1).contiguous()
    z["rs_in1"], z["rs_in2"] = i1.numpy(), i2.numpy()
    out = ref_ext.resample2d_fwd(i1.to(dev), i2.to(dev), 4, 1)
    z["rs_out"] = out.cpu().numpy()
    g = randn(tuple(out.shape), seed=133)
    g1, g2 = ref_ext.resample2d_bwd(i1.to(dev), i2.to(dev), g.to(dev), 4, 1)
    z["rs_gout"], z["rs_gin1"], z["rs_gin2"] = g.numpy(), g1.cpu().numpy(), g2.cpu().numpy()
    out2 = ref_ext.resample2d_fwd(i1.to(dev), i2.to(dev), 2, 1)
    z["rs_out_k2"] = out2.cpu().numpy()
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    np.savez_compressed(out_path, **z)
    print("wrote", out_path, {k: v.shape for k, v in z.items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_golden.npz"))
