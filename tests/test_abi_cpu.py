"""The C-ABI library loads and exports every symbol include/gfla_hip.h declares; host-side
argument checks behave like the reference's (no compute is launched here)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    """Expand the GFLA_DECL_* macros of the header by hand: every `gfla_<name>_##SFX` under a
    macro that is instantiated with (sfx, type)."""
    text = open(os.path.join(ROOT, "include", "gfla_hip.h")).read()
    names = set(re.findall(r"^(?:int|int64_t|const char \*)\s*(gfla_\w+)\(", text, flags=re.M))
    for macro, body in re.findall(r"#define (GFLA_DECL_\w+)\(SFX, T\)(.*?)\n(?=GFLA_DECL)", text, flags=re.S):
        bases = re.findall(r"(gfla_\w+?)_##SFX", body)
        for sfx in re.findall(macro + r"\((\w+), \w+\)", text):
            if sfx == "SFX":
                continue
            names.update("%s_%s" % (b, sfx) for b in bases)
    return names


def test_header_and_library_agree(gfla):
    from global_flow_local_attention_amd import _lib
    gfla.build()
    declared = _declared_symbols()
    assert declared == set(gfla.exported_symbols()), declared ^ set(gfla.exported_symbols())
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(handle, name), name
    assert _lib.lib().gfla_abi_version() == 1


def test_argument_validation_without_gpu(gfla):
    from global_flow_local_attention_amd import _lib
    L = _lib.lib()
    n = None
    assert L.gfla_block_extractor_fwd_f32(n, n, n, 1, 1, 4, 4, 4, 4, 3, n) == -1  # NULL pointer
    assert L.gfla_local_attn_reshape_fwd_f32(n, n, 1, 4, 4, 3, n) == -1
    buf = (ctypes.c_float * 4)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.gfla_block_extractor_fwd_f32(p, p, p, 0, 1, 4, 4, 4, 4, 3, n) == -2   # bad shape
    assert L.gfla_resample2d_fwd_f32(p, p, p, 1, 1, 4, 4, 4, 4, 1, 1, n) == -2     # kernel_size < 2
    assert L.gfla_local_attn_aggregate_fwd_f32(p, p, p, p, n, 1, 1, 4, 4, 4, 4, 9, 1, n) == -3  # k > 5
    assert b"NULL" in L.gfla_status_string(-1)


def test_cpu_tensors_are_rejected_like_the_reference(gfla):
    # block_extractor.py:23-24 / local_attn_reshape.py:20-21 raise NotImplementedError on CPU
    with pytest.raises(NotImplementedError):
        gfla.BlockExtractor(3)(torch.zeros(1, 1, 4, 4), torch.zeros(1, 2, 4, 4))
    with pytest.raises(NotImplementedError):
        gfla.LocalAttnReshape()(torch.zeros(1, 9, 4, 4), 3)
    with pytest.raises(NotImplementedError):
        gfla.Resample2d(4, 1, 2)(torch.zeros(1, 1, 4, 4), torch.zeros(1, 2, 4, 4))
    with pytest.raises(NotImplementedError):
        gfla.ExtractorAttn(4, 3, softmax=True)(torch.zeros(1, 4, 4, 4), torch.zeros(1, 4, 4, 4), torch.zeros(1, 2, 4, 4))


def test_module_surface_matches_reference(gfla):
    m = gfla.ExtractorAttn(16, 5, torch.nn.LeakyReLU(0.1), softmax=True)
    assert list(m.state_dict().keys()) == ["fully_connect_layer.0.weight", "fully_connect_layer.0.bias",
                                           "fully_connect_layer.2.weight", "fully_connect_layer.2.bias"]
    assert m.fully_connect_layer[0].weight.shape == (128, 32, 5, 5)
    assert m.fully_connect_layer[2].weight.shape == (25, 128, 1, 1)
    assert isinstance(m.fully_connect_layer[3], torch.nn.Softmax)
    assert isinstance(gfla.ExtractorAttn(16, 4).fully_connect_layer[3], torch.nn.LeakyReLU)  # softmax=None
    assert gfla.BlockExtractor().kernel_size == 3
    r = gfla.Resample2d()
    assert (r.kernel_size, r.dilation) == (2, 1)
    assert len(list(r.parameters())) == 0 and len(list(r.buffers())) == 0


@pytest.mark.skipif(not os.path.isdir("/root/reference/model/networks"), reason="reference checkout not present")
def test_install_into_unmodified_reference(gfla):
    import subprocess, sys
    code = r"""
import sys, types
sys.path.insert(0, %r)
import global_flow_local_attention_amd as g
sys.modules.setdefault('torchvision', types.ModuleType('torchvision'))
bf = g.install('/root/reference')
import model.networks.generator as gen
net = gen.PoseGenerator(image_nc=3, structure_nc=18, ngf=64, img_f=512, layers=3, num_blocks=2, use_spect=False,
                        attn_layer=[2, 3], norm='instance', activation='LeakyReLU', extractor_kz={'2': 5, '3': 3})
assert type(net.target.attn0.extractor) is g.BlockExtractor
assert type(net.target.attn0.reshape) is g.LocalAttnReshape
assert bf.ExtractorAttn.forward.__module__ == 'global_flow_local_attention_amd.extractor_attn'
assert abs(sum(p.numel() for p in net.parameters()) - 14047395) == 0
print('ok')
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
