"""MI355X (gfx950) implementation of the Global-Flow-Local-Attention feature-warping hot path.

Public surface = the reference's: BlockExtractor, LocalAttnReshape, Resample2d (+ their autograd
Functions) and ExtractorAttn; `install()` wires them into an unmodified reference checkout.
All compute is in libgfla_hip.so (csrc/, C ABI in include/gfla_hip.h); there is no CPU path.
"""
from ._lib import build, exported_symbols, path_count, set_tuning  # noqa: F401
from .block_extractor import BlockExtractor, BlockExtractorFunction  # noqa: F401
from .local_attn_reshape import LocalAttnReshape, LocalAttnReshapeFunction  # noqa: F401
from .resample2d import Resample2d, Resample2dFunction  # noqa: F401
from .extractor_attn import (BlockExtractorUnfoldFunction, ExtractorAttn, FcTailFunction, GraphedCall,  # noqa: F401
                             LocalAttnAggregateFunction, graphed_inference, patch_reference_extractor_attn)
from .losses import AffineRegularizationLoss, MultiAffineRegularizationLoss  # noqa: F401
from .correctness import CorrectnessMapFunction, MaxCosineFunction, PerceptualCorrectness, max_cosine_similarity  # noqa: F401
from .install import install  # noqa: F401
from .trainer import TrainerShell, load_reference_checkpoint  # noqa: F401
from .face_step import (DualStreamAttn, MaskBlendFunction, face_target_forward, generate_frames,  # noqa: F401
                        patch_reference_face_target_net)

__version__ = "0.1.0"
