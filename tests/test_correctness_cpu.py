"""Sampling-correctness loss: the host restatement (oracle/cpu_modules.py) against golden vectors made by
the reference's own `PerceptualCorrectness.calculate_loss` (tests/golden/make_correctness_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle.cpu_modules import PerceptualCorrectnessCPU, max_cosine_cpu

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "correctness_golden.npz")
CASES = ("c16_12x10", "c24_9x13_mask", "c64_16x11")


@pytest.fixture(scope="module")
def z():
    return np.load(PATH)


def case(z, name, device="cpu"):
    t = {k.split("/", 1)[1]: torch.from_numpy(z[k]).to(device) for k in z.files if k.startswith(name + "/")}
    t.setdefault("mask", None)
    return t


@pytest.mark.parametrize("name", CASES)
def test_oracle_loss_and_gradients_vs_reference_golden(oracle, z, name):
    g = case(z, name)
    src, tgt, flow = (g[k].clone().requires_grad_() for k in ("src", "tgt", "flow"))
    mod = PerceptualCorrectnessCPU()
    mod.target_vgg, mod.source_vgg = {"f": tgt}, {"f": src}
    loss = mod.calculate_loss(flow, "f", g["mask"])
    loss.backward()
    assert abs(loss.item() - g["loss"].item()) <= 1e-6
    for got, want in ((src.grad, g["g_src"]), (tgt.grad, g["g_tgt"]), (flow.grad, g["g_flow"])):
        assert (got - want).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("name", CASES)
def test_max_cosine_oracle_vs_reference_golden(z, name):
    g = case(z, name)
    B, C = g["src"].shape[:2]
    best, idx = max_cosine_cpu(g["src"].view(B, C, -1), g["tgt"].view(B, C, -1))
    assert torch.equal(idx, g["idx"])
    assert (best - g["best"]).abs().max().item() <= 1e-6


def test_max_cosine_entry_point_is_exported(gfla):
    assert "gfla_max_cosine_fwd_f32" in gfla.exported_symbols()
    assert hasattr(gfla._lib.lib(), "gfla_max_cosine_fwd_f32")


def test_max_cosine_rejects_host_tensors(gfla):
    with pytest.raises(NotImplementedError):
        gfla.max_cosine_similarity(torch.zeros(1, 4, 3, 3), torch.zeros(1, 4, 3, 3))
