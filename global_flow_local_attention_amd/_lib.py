"""ctypes binding of libgfla_hip.so (the C ABI declared in include/gfla_hip.h).

The library is the only implementation of the ops: there is no Python/torch fallback.  If it
is missing or fails to load, importing an op raises; if a call returns a non-zero status, a
RuntimeError is raised (the reference swallows native errors, block_extractor_cuda.cc:11).
"""
import ctypes
import os
import subprocess

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# GFLA_HIP_LIBRARY: a differently built library (tools/ubench/build_agg_abl.sh timing variants); default = the in-tree build
LIB_PATH = os.environ.get("GFLA_HIP_LIBRARY") or os.path.join(_PKG, "libgfla_hip.so")
_lib = None

_i64, _int, _ptr = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p

# entry point -> argument types (pointers first, then sizes), mirroring include/gfla_hip.h
_SIGNATURES = {
    "gfla_block_extractor_fwd": [_ptr] * 3 + [_i64] * 6 + [_int, _ptr],
    "gfla_block_extractor_bwd": [_ptr] * 5 + [_i64] * 6 + [_int, _ptr],
    "gfla_block_extractor_unfold_fwd": [_ptr] * 3 + [_i64] * 6 + [_int, _int, _ptr],
    "gfla_block_extractor_unfold_bwd": [_ptr] * 5 + [_i64] * 6 + [_int, _int, _ptr],
    "gfla_local_attn_reshape_fwd": [_ptr] * 2 + [_i64] * 3 + [_int, _ptr],
    "gfla_local_attn_reshape_bwd": [_ptr] * 2 + [_i64] * 3 + [_int, _ptr],
    "gfla_resample2d_fwd": [_ptr] * 3 + [_i64] * 6 + [_int, _int, _ptr],
    "gfla_resample2d_bwd": [_ptr] * 5 + [_i64] * 6 + [_int, _int, _int, _ptr],
    "gfla_local_attn_aggregate_fwd": [_ptr] * 5 + [_i64] * 6 + [_int, _int, _ptr],
    "gfla_local_attn_aggregate_bwd": [_ptr] * 7 + [_i64] * 6 + [_int, _int, _ptr],
    "gfla_local_attn_source_bwd": [_ptr] * 7 + [_i64] * 6 + [_int, _int, _ptr],
}
# entry points that exist in one precision only: full symbol name -> argument types
_SINGLE = {
    "gfla_max_cosine_fwd_f32": [_ptr] * 5 + [_i64] * 4 + [ctypes.c_double, _ptr],
    "gfla_max_cosine_workspace_bytes": [_i64] * 3,
    "gfla_fc_tail_fwd_f32": [_ptr, _i64, _i64] + [_ptr] * 5 + [_i64] * 3 + [_int, ctypes.c_double, _ptr],
    "gfla_fc_tail_fwd_f64": [_ptr, _i64, _i64] + [_ptr] * 5 + [_i64] * 3 + [_int, ctypes.c_double, _ptr],
    "gfla_fc_tail_bwd_f32": [_ptr, _i64, _i64] + [_ptr] * 8 + [_i64] * 3 + [_int, ctypes.c_double, _ptr],
    "gfla_fc_tail_bwd_f64": [_ptr, _i64, _i64] + [_ptr] * 8 + [_i64] * 3 + [_int, ctypes.c_double, _ptr],
    "gfla_replicate_pad_bwd_f32": [_ptr] * 2 + [_i64] * 3 + [_int] * 4 + [_ptr],
    "gfla_replicate_pad_bwd_f64": [_ptr] * 2 + [_i64] * 3 + [_int] * 4 + [_ptr],
    "gfla_correctness_map_fwd_f32": [_ptr] * 5 + [_i64] * 3 + [ctypes.c_double] * 2 + [_ptr],
    "gfla_correctness_map_bwd_f32": [_ptr] * 9 + [_i64] * 3 + [ctypes.c_double] * 2 + [_ptr],
    "gfla_fc_supported": [_i64] * 3 + [_int, _int],
    "gfla_fc_workspace_bytes": [_i64] * 4 + [_int] * 3,
    "gfla_fc_forward_f32": [_ptr] * 9 + [_i64] * 4 + [_int, ctypes.c_double, _int, _ptr],
    "gfla_fc_backward_f32": [_ptr] * 12 + [_i64] * 4 + [_int, ctypes.c_double, _int, _int, _ptr],
    "gfla_fc_geometry": [_i64, _i64, _int, _int, _ptr],
    "gfla_fc_conv_fwd_f32": [_ptr, _ptr, _int, _ptr, _ptr] + [_i64] * 4 + [_int, _int, _ptr],
    "gfla_fc_conv_bwd_f32": [_ptr, _int, _ptr, _ptr, _ptr, _ptr] + [_i64] * 4 + [_int, _int, _ptr],
    "gfla_fc_tr_probe": [_ptr, _int, _ptr, _ptr, _ptr],
    "gfla_fc_wino_debug_buffer": [_ptr],
    "gfla_fc_kernel_f32": [_int, _ptr, _ptr] + [_i64] * 4 + [_int, _int, _ptr],
    "gfla_scatter_workspace_bytes": [_i64] * 3 + [_int],
    "gfla_aggregate_fwd_workspace_bytes": [_i64] * 3 + [_int],
    "gfla_aggregate_fwd_geometry": [_i64] * 6 + [_int, _ptr],
    "gfla_aggregate_bwd_supported": [_i64, _i64, _int],
    "gfla_big_plane_geometry": [_int] + [_i64] * 6 + [_int, _int, _ptr],
    "gfla_xcd_swizzle": [_i64, _i64],
    "gfla_local_attn_aggregate_fwd_ws_f32": [_ptr] * 6 + [_i64] * 6 + [_int, _int, _ptr],
    "gfla_local_attn_aggregate_fwd_ws_bf16": [_ptr] * 6 + [_i64] * 6 + [_int, _int, _ptr],
    "gfla_local_attn_aggregate_bwd_ws_f32": [_ptr] * 8 + [_i64] * 6 + [_int, _int, _ptr],
    "gfla_resample2d_bwd_ws_f32": [_ptr] * 6 + [_i64] * 6 + [_int, _int, _int, _ptr],
    "gfla_convert_multi": [_ptr, _ptr, _i64] * 4 + [_int, _ptr],
    "gfla_mask_blend_fwd_f32": [_ptr] * 6 + [_i64] * 3 + [_ptr],
    "gfla_mask_blend_fwd_bf16": [_ptr] * 6 + [_i64] * 3 + [_ptr],
    "gfla_mask_blend_bwd_f32": [_ptr] * 11 + [_i64] * 3 + [_ptr],
    "gfla_mask_blend_bwd_bf16": [_ptr] * 11 + [_i64] * 3 + [_ptr],
}
# bf16 storage exists for every entry point below; the backward ones return the reductions over channels (grad_flow,
# grad_logits, grad_in2) in float32 (include/gfla_hip.h)
_FWD_ONLY_BF16 = set()


def exported_symbols():
    """Every symbol include/gfla_hip.h declares."""
    names = ["gfla_abi_version", "gfla_status_string", "gfla_set_tuning", "gfla_path_count", "gfla_unfold_supported"]
    for base in _SIGNATURES:
        for sfx in ("f32", "f64", "bf16"):
            if sfx == "bf16" and base in _FWD_ONLY_BF16:
                continue
            names.append("%s_%s" % (base, sfx))
    return names + list(_SINGLE)


def build(force=False):
    """Compile csrc/*.hip for gfx950 into libgfla_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_PKG, "csrc"), "-j8"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libgfla_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C global_flow_local_attention_amd/csrc`. There is no fallback path." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        handle.gfla_status_string.restype = ctypes.c_char_p
        handle.gfla_status_string.argtypes = [_int]
        handle.gfla_set_tuning.argtypes = [_int, _int]
        handle.gfla_path_count.argtypes = [_int]
        handle.gfla_path_count.restype = _i64
        handle.gfla_unfold_supported.argtypes = [_i64, _i64, _int, _int]
        for base, args in _SIGNATURES.items():
            for sfx in ("f32", "f64", "bf16"):
                if sfx == "bf16" and base in _FWD_ONLY_BF16:
                    continue
                fn = getattr(handle, "%s_%s" % (base, sfx))
                fn.argtypes = args
                fn.restype = _int
        for name, args in _SINGLE.items():
            fn = getattr(handle, name)
            fn.argtypes = args
            fn.restype = _i64 if (name.endswith("_bytes") or name == "gfla_xcd_swizzle") else _int
        _lib = handle
    return _lib


_SUFFIX = {torch.float32: "f32", torch.float64: "f64", torch.bfloat16: "bf16"}


def suffix(t, what, allow_bf16=True):
    """Entry-point suffix for t's dtype.  allow_bf16=False: entry points without bfloat16 storage raise a clear
    TypeError instead of a missing-symbol AttributeError."""
    try:
        sfx = _SUFFIX[t.dtype]
    except KeyError:
        raise TypeError("%s: unsupported dtype %s (float32, float64, bfloat16 forward)" % (what, t.dtype))
    if sfx == "bf16" and not allow_bf16:
        raise TypeError("%s: bfloat16 is forward-only in this library (use float32 for training)" % what)
    return sfx


def reduction_like(t):
    """Zeroed buffer for a gradient that is a reduction over channels (grad_flow, grad_logits, grad_in2): float32 when the
    storage type is bfloat16 (the bf16 backward entry points accumulate these in float32), t's dtype otherwise."""
    return torch.zeros(t.shape, dtype=torch.float32 if t.dtype == torch.bfloat16 else t.dtype, device=t.device)


def scatter_workspace(ref_tensor, B, H, W, entries):
    """Scratch for the matrix-core scatter paths (csrc/patch_mfma.hip): the patch table of one op invocation."""
    n = lib().gfla_scatter_workspace_bytes(int(B), int(H), int(W), int(entries))
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=ref_tensor.device)


def aggregate_fwd(source, flow, logits, out, attn, k, apply_softmax):
    """softmax + aggregate forward.  f32 / bf16 storage: the coefficient-table kernels (scratch from the caching
    allocator, csrc/local_attn_aggregate.hip); f64: the plain entry point."""
    b, c, hs, ws = source.shape
    h, w = flow.shape[2], flow.shape[3]
    sfx = suffix(source, "local_attn_aggregate")
    tail = (b, c, hs, ws, h, w, int(k), 1 if apply_softmax else 0)
    if sfx in ("f32", "bf16"):
        n = lib().gfla_aggregate_fwd_workspace_bytes(int(b), int(h), int(w), int(k))
        scratch = torch.empty(max(int(n), 16), dtype=torch.uint8, device=source.device)
        call("gfla_local_attn_aggregate_fwd_ws_" + sfx, source, ptr(source), ptr(flow), ptr(logits), ptr(out), ptr(attn),
             ptr(scratch), *tail)
    else:
        call("gfla_local_attn_aggregate_fwd_" + sfx, source, ptr(source), ptr(flow), ptr(logits), ptr(out), ptr(attn), *tail)


def require_gpu(*tensors):
    """The reference raises NotImplementedError for non-CUDA tensors (block_extractor.py:23-24)."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NotImplementedError("GFLA ops run on the GPU only (got a %s tensor); there is no CPU path"
                                      % t.device.type)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Unsupported(RuntimeError):
    """GFLA_ERR_UNSUPPORTED (-3): the arguments are valid but outside what this entry point's kernels take (nothing was
    launched).  Callers that have another way to the same result catch exactly this."""


def call(name, ref_tensor, *args):
    """Invoke `name` on the current stream of ref_tensor's device; raise on non-zero status."""
    fn = getattr(lib(), name)
    with torch.cuda.device(ref_tensor.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(ref_tensor.device).cuda_stream)
        status = fn(*args, stream)
    if status != 0:
        err = Unsupported if status == -3 else RuntimeError
        raise err("%s failed: %s (status %d)" % (name, lib().gfla_status_string(status).decode(), status))


def convert_many(tensors, dtype):
    """[t.to(dtype) for t in tensors] for bfloat16 <-> float32 CUDA tensors, up to four per launch (gfla_convert_multi);
    None entries pass through.  Anything else (other dtypes, CPU tensors, nothing to convert) goes to torch."""
    out = list(tensors)
    todo = [i for i, t in enumerate(out) if t is not None and t.dtype != dtype]
    pair_ok = all(out[i].is_cuda and {out[i].dtype, dtype} == {torch.bfloat16, torch.float32} for i in todo)
    if not todo or not pair_ok or len({out[i].dtype for i in todo}) != 1:
        return [None if t is None else t.to(dtype) for t in out]
    for at in range(0, len(todo), 4):
        grp = todo[at:at + 4]
        srcs = [out[i].contiguous() for i in grp]
        dsts = [torch.empty(t.shape, dtype=dtype, device=t.device) for t in srcs]
        args = []
        for j in range(4):
            args += [ptr(srcs[j]), ptr(dsts[j]), srcs[j].numel()] if j < len(grp) else [None, None, 0]
        call("gfla_convert_multi", srcs[0], *args, 1 if dtype == torch.bfloat16 else 0)
        for i, d in zip(grp, dsts):
            out[i] = d
    return out


def unfold_supported(Hs, Ws, k, elem_size):
    return bool(lib().gfla_unfold_supported(int(Hs), int(Ws), int(k), int(elem_size)))


def set_tuning(key, value):
    """Process-global tuning knob (include/gfla_hip.h); returns the old value."""
    return lib().gfla_set_tuning(int(key), int(value))


ABI_VERSION = 8
# dispatch-trace ids (enum gfla_path in include/gfla_hip.h)
PATH_BE_BWD_LDS, PATH_BE_BWD_GLOBAL, PATH_FC_FWD_MODE0, PATH_FC_BWD_MODE0, PATH_BE_FWD_PIX = 0, 1, 2, 7, 12
# round 5: the big-plane kernels (few planes, each beyond the LDS budget; csrc/tile_map.h)
PATH_BE_FWD_GPIX, PATH_BE_BWD_TILE, PATH_RS_FWD_BIG, PATH_RS_BWD1_TILE, PATH_RS_BWD2_BIG = 13, 14, 15, 16, 17
PATH_FC_FWD_MODE5, PATH_FC_BWD_MODE5, PATH_COUNT = 18, 19, 20


def fc_path(mode, backward=False):
    """Dispatch-trace id of gfla_fc_forward_f32 / gfla_fc_backward_f32 in arithmetic mode `mode` (modes 0-4: ids 2-6 / 7-11)."""
    if int(mode) == 5:
        return PATH_FC_BWD_MODE5 if backward else PATH_FC_FWD_MODE5
    return (PATH_FC_BWD_MODE0 if backward else PATH_FC_FWD_MODE0) + int(mode)


def path_count(path):
    """How many times kernel path `path` has been enqueued by this process (any host thread)."""
    return int(lib().gfla_path_count(int(path)))
