"""Seeded synthetic inputs shared by the CPU and GPU tests."""
import torch
import torch.nn.functional as F

FLOW_KINDS = ("zero", "coherent", "wild", "smooth", "integer")


def make_flow(kind, B, H, W, dtype=torch.float32, seed=0):
    """(B,2,H,W) flow in source pixels.
    zero: identity (pure unfold); coherent: randn*2; wild: randn*8 (reaches out of bounds);
    smooth: low-pass noise (closest to real flow fields); integer: exact integers (weights 0/1)."""
    g = torch.Generator().manual_seed(seed)
    if kind == "zero":
        return torch.zeros(B, 2, H, W, dtype=dtype)
    n = torch.randn(B, 2, H, W, generator=g, dtype=torch.float64)
    if kind == "coherent":
        f = n * 2
    elif kind == "wild":
        f = n * 8
    elif kind == "smooth":
        f = F.avg_pool2d(F.pad(n * 12, (3, 3, 3, 3), mode="replicate"), 7, 1)
    elif kind == "integer":
        f = torch.round(n * 3)
    else:
        raise ValueError(kind)
    return f.to(dtype).contiguous()


def randn(shape, dtype=torch.float32, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64).to(dtype).contiguous()


def rand(shape, dtype=torch.float32, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g, dtype=torch.float64).to(dtype).contiguous()


def max_abs(a, b):
    return (a.double() - b.double()).abs().max().item()


def assert_close(got, want, tol, what=""):
    """max|got-want| <= tol * max(1, max|want|): absolute for O(1) outputs, relative for the
    large accumulated gradients."""
    scale = max(1.0, want.double().abs().max().item())
    err = max_abs(got, want)
    assert err <= tol * scale, "%s: max abs err %.3e > %.1e * %.3g" % (what, err, tol, scale)
    return err
