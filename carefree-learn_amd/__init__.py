"""carefree-learn_amd — MI355X-native (gfx950) implementation of carefree-learn's data-parallel
training hot path: the `cflearn.modules` building blocks behind ViT (Linear / Attention /
FeedForward / LayerNorm / patch embedding), the conv -> BatchNorm -> LeakyReLU stack of the MNIST
classifier and the FCNN head (Conv2d as im2row + MFMA GEMM), the fused Adam step and the bucketed
RCCL gradient exchange, behind the reference's module-registry surface.

Importable as `cflearn_amd` (the directory name `carefree-learn_amd` is not a Python identifier;
`cflearn_amd.py` at the repository root is the alias loader).

Importing the package never touches the GPU; the first kernel call loads `libcfhip.so`
(built by `__graft_entry__.build()`), and raises if it is missing — there is no CPU fallback.
"""
from . import _lib, ops, functional, fused, registry, modules, optim, ddp, contrastive, checkpoint  # noqa: F401
from .checkpoint import AsyncCheckpointer  # noqa: F401
from .registry import (  # noqa: F401
    PrefixModules,
    build_module,
    module_dict,
    override_reference_registry,
    register_module,
)
from .modules import (  # noqa: F401
    Attention,
    AttentionTokenMixer,
    BatchNorm1d,
    BatchNorm2d,
    Conv2d,
    FCNN,
    FeedForward,
    HijackCustomLinear,
    HijackLinear,
    LayerNorm,
    Linear,
    MixedStackedEncoder,
    MixingBlock,
    Mapping,
    NormFactory,
    VanillaClassifier,
    VanillaEncoder,
    VanillaEncoder1D,
    VanillaPatchEmbed,
    ViTEncoder,
    vit_b16_classifier,
)
from .optim import FusedAdam, FusedAdamOptimizer, FusedAdamWOptimizer, ParamArena, clip_grad_norm_  # noqa: F401
from .ddp import BucketedAllReduce, RcclDDPCallback, get_ddp_info  # noqa: F401

__version__ = "0.1.0"
