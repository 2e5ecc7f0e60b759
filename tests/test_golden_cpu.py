"""The CPU oracle against golden vectors produced by the REAL reference kernels
(tests/golden/ref_golden.npz, made by tests/golden/make_ref_golden.py on an MI355X from the
unmodified reference sources).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from util import assert_close

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_golden.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="ref_golden.npz not generated yet")


@pytest.fixture(scope="module")
def z():
    data = np.load(PATH)
    return {k: torch.from_numpy(data[k]) for k in data.files}


def test_block_extractor_oracle_vs_reference_golden(oracle, z):
    for k in (3, 5):
        out = oracle.block_extractor_fwd(z["be_source"], z["be_flow"], k)
        assert_close(out, z["be_out_k%d" % k], 2e-6, "fwd k=%d" % k)
        gs, gf = oracle.block_extractor_bwd(z["be_source"], z["be_flow"], z["be_gout_k%d" % k], k)
        assert_close(gs, z["be_gsrc_k%d" % k], 1e-5, "grad_source k=%d" % k)
        assert_close(gf, z["be_gflow_k%d" % k], 1e-5, "grad_flow k=%d" % k)
    assert_close(oracle.block_extractor_fwd(z["be2_source"], z["be2_flow"], 3), z["be2_out_k3"], 2e-6, "Hs != Hf")
    assert_close(oracle.block_extractor_fwd(z["be3_source"], z["be3_flow"], 3), z["be3_out_k3"], 1e-13, "fp64")


def test_local_attn_reshape_oracle_vs_reference_golden(oracle, z):
    assert torch.equal(oracle.local_attn_reshape_fwd(z["lar_in"], 3), z["lar_out"])
    assert torch.equal(oracle.local_attn_reshape_bwd(z["lar_gout"], 3), z["lar_gin"])


def test_resample2d_oracle_vs_reference_golden(oracle, z):
    assert_close(oracle.resample2d_fwd(z["rs_in1"], z["rs_in2"], 4, 1), z["rs_out"], 4e-6, "fwd k=4")
    assert_close(oracle.resample2d_fwd(z["rs_in1"], z["rs_in2"], 2, 1), z["rs_out_k2"], 4e-6, "fwd k=2")
    g1, g2 = oracle.resample2d_bwd(z["rs_in1"], z["rs_in2"], z["rs_gout"], 4, 1, trunc_compat=True)
    assert_close(g1, z["rs_gin1"], 1e-5, "grad_input1 incl. the int() truncation quirk")
    assert_close(g2, z["rs_gin2"], 1e-4, "grad_input2 (dx,dy,sigma)")
    # and the floor variant must NOT match the reference here (inputs reach negative coordinates)
    g1f, _ = oracle.resample2d_bwd(z["rs_in1"], z["rs_in2"], z["rs_gout"], 4, 1, trunc_compat=False)
    assert (g1f - z["rs_gin1"]).abs().max() > 1e-3
