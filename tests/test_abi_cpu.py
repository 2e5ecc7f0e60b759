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
    assert _lib.lib().gfla_abi_version() == _lib.ABI_VERSION == 8


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


def test_aggregate_forward_table_path_geometry_invariants():
    """Host-side launch logic of the coefficient-table aggregation forward (csrc/local_attn_aggregate.hip), swept over
    shapes without a GPU: whatever it picks has to fit the hardware (160 KB of LDS, 12 waves per workgroup), cover every
    channel and tile, keep the staging within its per-thread budget of word pairs, and keep the LDS pitch on the
    conflict-free residue; the scratch the size query promises has to hold a record for every tile slot."""
    from global_flow_local_attention_amd import _lib
    L = _lib.lib()
    out = (ctypes.c_int64 * 9)()
    seen = 0
    for B in (1, 2, 8, 32, 256):
        for C in (1, 3, 16, 64, 128, 256, 512):
            for (H, W) in ((8, 8), (32, 22), (64, 44), (64, 64), (32, 32), (128, 88), (256, 176), (7, 6), (9, 130)):
                for k in (1, 3, 5):
                    rc = L.gfla_aggregate_fwd_geometry(B, C, H, W, H, W, k, ctypes.cast(out, ctypes.c_void_p))
                    if W < k + 1 or W % 2:
                        assert rc == -3
                        continue
                    if rc == -3:      # plane too large for two LDS buffers: the plain kernels take it
                        assert H * W * 4 * 2 > 60 * 1024
                        continue
                    assert rc == 0, (B, C, H, W, k, rc)
                    CH, CS, ns, tg, threads, pitch, tw, ntile, lds = list(out)
                    seen += 1
                    assert 1 <= CH <= C and CS % CH == 0 and ns * CS >= C and (ns - 1) * CS < C
                    assert threads % 64 == 0 and 64 <= threads <= 12 * 64
                    assert tw in (8, 16, 32) and ntile == -(-W // tw) * -(-H // (64 // tw))
                    assert tg * (threads // 64) >= ntile                    # every tile has a wave
                    assert pitch >= 2 * W and pitch % 64 == 32
                    per_plane = -(-H // 2) * pitch * 4
                    assert lds == 2 * (CH * per_plane + 16) and lds <= 160 * 1024
                    assert CH * H * (W // 2) <= 8 * threads                 # word pairs per thread and chunk
                    assert L.gfla_aggregate_fwd_workspace_bytes(B, H, W, k) >= B * ntile * 64 * (((k + 1) * (k + 2) + 4) // 4 * 4) * 4
    assert seen > 300


def test_host_helpers_of_the_launch_merges(gfla):
    """Round 4's host-side helpers, on CPU tensors: the one-allocation zero arena of the attention backward (segments on
    256-byte boundaries, the requested shapes, None for gradients nobody wants) and convert_many's pass-through rules
    (anything but a bf16 <-> f32 CUDA conversion goes to torch; None stays None; nothing to convert = the same tensors)."""
    import torch
    from global_flow_local_attention_amd import _lib, extractor_attn
    a, b, c = extractor_attn._zeros_f32(torch.device("cpu"), ((2, 3, 5, 7), True), ((2, 2, 5, 7), False), ((2, 9, 5, 7), True))
    assert b is None and a.shape == (2, 3, 5, 7) and c.shape == (2, 9, 5, 7)
    assert a.dtype == c.dtype == torch.float32 and not a.any() and not c.any()
    assert a.is_contiguous() and c.is_contiguous()
    assert a.untyped_storage().data_ptr() == c.untyped_storage().data_ptr()          # one allocation ...
    assert (c.data_ptr() - a.data_ptr()) % 256 == 0 and c.data_ptr() - a.data_ptr() >= a.numel() * 4   # ... aligned segments
    a.fill_(1.0)
    assert not c.any()
    x32, x16 = torch.randn(5, 3), torch.randn(4).to(torch.bfloat16)
    out = _lib.convert_many([x32, None, x16], torch.float32)
    assert out[0] is x32 or torch.equal(out[0], x32)
    assert out[1] is None and out[2].dtype == torch.float32 and torch.equal(out[2], x16.float())
    back = _lib.convert_many([x32], torch.bfloat16)
    assert back[0].dtype == torch.bfloat16 and torch.equal(back[0], x32.to(torch.bfloat16))


def test_big_plane_tile_geometry_and_xcd_remap_invariants(gfla):
    """Host logic of round 5's tile kernels (csrc/tile_map.h) without a GPU.  Geometry: whatever it picks must fit the
    hardware (<= 512 threads in whole waves, one pixel per thread, LDS request within the budget and at least 16 KB), cover
    the map with its tiles and the channels with its groups, start tiles on 32-column boundaries, and leave at least three
    workgroups per CU when the problem has them.  Regime: BASELINE configs[1] is in, the bench shapes (planes in LDS) and
    many-plane problems are out.  Remap: a bijection of [0, nwg) for any nwg, and the blocks of one XCD (block % 8) get one
    contiguous range."""
    from global_flow_local_attention_amd import _lib
    L = _lib.lib()
    out = (ctypes.c_int64 * 10)()
    po = ctypes.cast(out, ctypes.c_void_p)
    seen = 0
    for op in (0, 1, 2, 3, 4):
        for B in (1, 2, 32):
            for C in (1, 3, 64, 256):
                for (H, W) in ((256, 176), (7, 5), (64, 44), (33, 600), (1, 1), (500, 31)):
                    for span in (3, 4, 6, 9):
                        assert L.gfla_big_plane_geometry(op, B, C, H, W, H, W, span, 4, po) == 0
                        regime, th, tw, ntx, nty, threads, G, ngroups, lds, nwg = list(out)
                        seen += 1
                        assert threads % 64 == 0 and th * tw <= threads <= 512 and th >= 1 and tw >= 1
                        assert ntx * tw >= W and (ntx - 1) * tw < W and nty * th >= H and (nty - 1) * th < H
                        assert tw == min(16 if op == 2 else 32, W)              # tiles start on 128-byte lines (resample2d forward: 64)
                        assert 1 <= G <= C and ngroups * G >= C and (ngroups - 1) * G < C
                        assert nwg == B * ntx * nty * ngroups
                        assert 16 * 1024 <= lds <= 64 * 1024 and lds % 256 == 0
                        if B * ntx * nty * C >= 3 * 256:                        # enough work: at least three workgroups per CU
                            assert nwg >= 3 * 256 or G == 1, (op, B, C, H, W, G, nwg)
    assert seen > 1000
    assert L.gfla_big_plane_geometry(0, 1, 64, 256, 176, 256, 176, 4, 4, po) == 0 and out[0] == 1     # configs[1]
    assert L.gfla_big_plane_geometry(3, 1, 64, 256, 176, 256, 176, 4, 4, po) == 0 and out[0] == 1
    assert L.gfla_big_plane_geometry(0, 32, 128, 64, 44, 64, 44, 6, 4, po) == 0 and out[0] == 0      # bench shape: planes in LDS
    assert L.gfla_big_plane_geometry(2, 32, 64, 256, 176, 256, 176, 4, 4, po) == 0 and out[0] == 0   # many planes: row windows
    assert L.gfla_big_plane_geometry(5, 1, 1, 1, 1, 1, 1, 1, 4, po) == -2 and L.gfla_big_plane_geometry(0, 1, 1, 1, 1, 1, 1, 1, 4, None) == -1
    for nwg in (1, 7, 8, 9, 63, 64, 65, 1536, 1537, 2815, 4099):
        img = [L.gfla_xcd_swizzle(b, nwg) for b in range(nwg)]
        assert sorted(img) == list(range(nwg)), nwg
        for x in range(min(8, nwg)):
            mine = [img[b] for b in range(x, nwg, 8)]
            assert mine == list(range(mine[0], mine[0] + len(mine))), (nwg, x)      # contiguous, in launch order
        assert L.gfla_xcd_swizzle(nwg, nwg) == -1 and L.gfla_xcd_swizzle(-1, nwg) == -1
