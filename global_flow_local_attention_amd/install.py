"""Make the UNMODIFIED reference network code consume these ops.

The reference imports its ops by path (base_function.py:10,12,13; external_function.py:5-7;
generator.py:6):
    model.networks.block_extractor.block_extractor          -> BlockExtractor
    model.networks.local_attn_reshape.local_attn_reshape    -> LocalAttnReshape
    model.networks.resample2d_package.resample2d            -> Resample2d
`install()` registers this package's modules under those names in sys.modules (plus light stubs
for packages the reference imports at module scope but that the hot path never calls), so that
`import model.networks.generator` from a reference checkout picks them up with no source change.
See INTEGRATION.md.
"""
import importlib
import os
import sys
import types

from . import block_extractor, local_attn_reshape, resample2d
from .extractor_attn import patch_reference_extractor_attn

_ALIASES = {
    "model.networks.block_extractor.block_extractor": block_extractor,
    "model.networks.local_attn_reshape.local_attn_reshape": local_attn_reshape,
    "model.networks.resample2d_package.resample2d": resample2d,
}


def _namespace(name, path=None):
    mod = sys.modules.get(name)
    if mod is None:
        mod = types.ModuleType(name)
        mod.__path__ = [path] if path else []
        sys.modules[name] = mod
    return mod


def install(reference_root=None, fuse_extractor_attn=True, stub_missing=True, allow_vendor_fallback=False,
            dual_stream_face=False):
    """Alias the three op modules; optionally patch the reference's ExtractorAttn with the fused
    forward.  `reference_root` (a checkout of the reference) is only needed if `model` is not
    already importable.  Returns the reference's `model.networks.base_function` module when it
    could be imported, else None.

    allow_vendor_fallback: an ExtractorAttn configuration this library's own MFMA kernels do not take (kernel_size other
    than 3 / 5 -- the reference's constructor default is 4 --, float64 features, maps too large for the LDS tiles) would
    run its FC layers through rocBLAS / MIOpen.  After install() that RAISES (extractor_attn.VendorFallbackError) unless
    this flag is True (then it warns once per module); the production configurations (kernel_size 2=5, 3=3) never reach it.

    dual_stream_face: also patch the reference's FaceTargetNet.forward (generator.py:480-505) so that the two ExtractorAttn
    of an attention layer (previous frame / reference frame) run on two HIP streams (face_step.py).  Imports the
    reference's generator module."""
    from . import extractor_attn as _ea
    _ea.VENDOR_FALLBACK = "warn" if allow_vendor_fallback else "error"
    if reference_root:
        # a bare namespace for `model` skips model/__init__.py (which pulls in skimage etc.)
        _namespace("model", os.path.join(reference_root, "model"))
        _namespace("model.networks", os.path.join(reference_root, "model", "networks"))
        if reference_root not in sys.path:
            sys.path.insert(0, reference_root)
    for pkg in ("model.networks.block_extractor", "model.networks.local_attn_reshape",
                "model.networks.resample2d_package"):
        if reference_root or pkg.rsplit(".", 1)[0] in sys.modules:
            _namespace(pkg)
    for name, mod in _ALIASES.items():
        sys.modules[name] = mod
        parent, _, leaf = name.rpartition(".")
        if parent in sys.modules:
            setattr(sys.modules[parent], leaf, mod)
    if stub_missing:
        for missing in ("imageio", "natsort"):
            try:
                importlib.import_module(missing)
            except ImportError:
                stub = types.ModuleType(missing)
                if missing == "natsort":
                    stub.natsorted = sorted
                sys.modules[missing] = stub
    base_function = None
    if "model.networks" in sys.modules:
        try:
            base_function = importlib.import_module("model.networks.base_function")
        except ImportError:
            base_function = None
    if base_function is not None and fuse_extractor_attn and hasattr(base_function, "ExtractorAttn"):
        patch_reference_extractor_attn(base_function.ExtractorAttn)
    if base_function is not None and dual_stream_face:
        from .face_step import patch_reference_face_target_net
        generator = importlib.import_module("model.networks.generator")
        patch_reference_face_target_net(generator.FaceTargetNet)
    return base_function
