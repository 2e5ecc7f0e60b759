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


def install(reference_root=None, fuse_extractor_attn=True, stub_missing=True, allow_vendor_fallback=None,
            dual_stream_face=False, strict_mfma=None):
    """Alias the three op modules; optionally patch the reference's ExtractorAttn with the fused
    forward.  `reference_root` (a checkout of the reference) is only needed if `model` is not
    already importable.  Returns the reference's `model.networks.base_function` module when it
    could be imported, else None.

    strict_mfma: an ExtractorAttn configuration this library's own MFMA kernels do not take (kernel_size other than
    3 / 5 -- the reference's constructor default is 4 --, float64 features and gradcheck, maps too large for the LDS tiles)
    runs its FC layers through rocBLAS / MIOpen, with ONE warning per module, and every such call is counted
    (extractor_attn.vendor_fallback_calls): a drop-in must not turn a working reference configuration into a failure.
    strict_mfma=True (or GFLA_STRICT_MFMA=1 in the environment) makes it RAISE extractor_attn.VendorFallbackError instead
    -- what a benchmark or a deployment that must not ship vendor kernels by accident wants (bench.py sets it).  None
    leaves the process-wide policy as it is, so a second install() never flips it silently.  allow_vendor_fallback is the
    round-4 spelling: True = "warn", False = "error".

    dual_stream_face: also patch the reference's FaceTargetNet.forward (generator.py:480-505) so that the two ExtractorAttn
    of an attention layer (previous frame / reference frame) run on two HIP streams (face_step.py).  Imports the
    reference's generator module."""
    from . import extractor_attn as _ea
    if strict_mfma is None and allow_vendor_fallback is not None:
        strict_mfma = not allow_vendor_fallback
    if strict_mfma is None and os.environ.get("GFLA_STRICT_MFMA"):
        strict_mfma = os.environ["GFLA_STRICT_MFMA"] not in ("0", "")
    if strict_mfma is not None:
        _ea.VENDOR_FALLBACK = "error" if strict_mfma else "warn"
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
