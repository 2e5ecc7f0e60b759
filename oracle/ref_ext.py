"""TEST INFRASTRUCTURE ONLY -- loader for the REAL reference extensions in oracle/_ref/.

oracle/build_ref.sh compiles the reference's own `*_cuda.cc` + `*_kernel.cu` (unmodified, in
place under /root/reference) into `block_extractor_cuda`, `local_attn_reshape_cuda` and
`resample2d_cuda` -- the same three pybind modules the reference imports
(block_extractor.py:3, local_attn_reshape.py:3, resample2d.py:4).  They need a GPU to run.

The helpers below do what the reference's autograd Functions do around those calls
(allocate zeroed outputs, pass saved tensors; block_extractor.py:8-42,
local_attn_reshape.py:8-37, resample2d.py:9-39) so tests can call the reference kernels
without the reference's Python files, which do not travel to the GPU box.
"""
import importlib
import os
import sys

import torch  # noqa: F401  (libtorch must be loaded before the extensions)

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
_MODS = {}


def available():
    return all(os.path.exists(os.path.join(_DIR, m + ".so"))
               for m in ("block_extractor_cuda", "local_attn_reshape_cuda", "resample2d_cuda"))


def _mod(name):
    if name not in _MODS:
        if _DIR not in sys.path:
            sys.path.insert(0, _DIR)
        _MODS[name] = importlib.import_module(name)
    return _MODS[name]


def block_extractor_fwd(source, flow, k):
    out = flow.new_zeros(source.size(0), source.size(1), k * flow.size(2), k * flow.size(3))
    _mod("block_extractor_cuda").forward(source, flow, out, k)
    return out


def block_extractor_bwd(source, flow, grad_out, k):
    gs, gf = torch.zeros_like(source), torch.zeros_like(flow)
    go = grad_out.contiguous()
    _mod("block_extractor_cuda").backward(source, flow, go, gs, gf, k)
    return gs, gf


def local_attn_reshape_fwd(inputs, k):
    out = inputs.new_zeros(inputs.size(0), 1, k * inputs.size(2), k * inputs.size(3))
    _mod("local_attn_reshape_cuda").forward(inputs, out, k)
    return out


def local_attn_reshape_bwd(inputs, grad_out, k):
    gi = torch.zeros_like(inputs)
    go = grad_out.contiguous()
    _mod("local_attn_reshape_cuda").backward(inputs, go, gi, k)
    return gi


def resample2d_fwd(input1, input2, k=2, dilation=1):
    out = input1.new_zeros(input2.size(0), input1.size(1), input2.size(2), input2.size(3))
    _mod("resample2d_cuda").forward(input1, input2, out, k, dilation)
    return out


def resample2d_bwd(input1, input2, grad_out, k=2, dilation=1):
    g1, g2 = torch.zeros_like(input1), torch.zeros_like(input2)
    go = grad_out.contiguous()
    _mod("resample2d_cuda").backward(input1, input2, go, g1, g2, k, dilation)
    return g1, g2
