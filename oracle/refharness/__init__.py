"""Import the reference's OWN hot-path modules from /root/reference (read-only).

TEST INFRASTRUCTURE ONLY (build container only: /root/reference does not exist on the GPU box).
Used by `oracle/gen_golden.py` and by the `not gpu` tests that pin `oracle/vit_oracle.py`
against the reference; never by the product path.

Recipe: package *shells* whose `__path__` points at the reference directories are
put in `sys.modules`, which skips the reference's `__init__.py` files (they pull in data / api /
torchvision), and a small `cftool` stand-in (./cftool) satisfies the un-vendored dependency.
"""
import importlib
import os
import sys
import types
from typing import Optional

REFERENCE_ROOT = os.environ.get("CFLEARN_REFERENCE_ROOT", "/root/reference")

_SHELLS = [
    "cflearn",
    "cflearn.modules",
    "cflearn.modules.cv",
    "cflearn.modules.cv.encoder",
    "cflearn.modules.cv.classifier",
    "cflearn.modules.ml",
    "cflearn.modules.nlp",
    "cflearn.modules.nlp.encoder",
    "cflearn.modules.multimodal",
    "cflearn.modules.multimodal.diffusion",
]

_loaded: Optional[types.SimpleNamespace] = None


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "cflearn", "modules", "core"))


def load_reference() -> types.SimpleNamespace:
    """Returns a namespace with the reference classes used by the oracle checks."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    here = os.path.dirname(os.path.abspath(__file__))
    if "cftool" not in sys.modules:
        sys.path.insert(0, here)
        importlib.import_module("cftool")
        for sub in ("misc", "array", "types", "pipeline", "cv"):
            importlib.import_module(f"cftool.{sub}")
    if "torchvision" not in sys.modules:
        # torchvision is not installed; the reference CLIP module imports transform class names at module
        # level (used only by `get_transform`, which the oracle never calls)
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        for name in ("Resize", "Compose", "ToTensor", "Normalize", "CenterCrop"):
            setattr(tvt, name, type(name, (), {"__init__": lambda self, *a, **k: None}))
        tvt.InterpolationMode = types.SimpleNamespace(BICUBIC="bicubic")
        tv.transforms = tvt
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt
    for name in _SHELLS:
        if name in sys.modules:
            continue
        shell = types.ModuleType(name)
        shell.__path__ = [os.path.join(REFERENCE_ROOT, *name.split("."))]  # type: ignore
        shell.__package__ = name
        sys.modules[name] = shell
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, shell)

    common = importlib.import_module("cflearn.modules.common")
    core = importlib.import_module("cflearn.modules.core")
    toolkit = importlib.import_module("cflearn.toolkit")
    # emulate `from .common import *` / `from .core import *` of cflearn/modules/__init__.py
    modules_shell = sys.modules["cflearn.modules"]
    for mod in (common, core):
        for k in dir(mod):
            if not k.startswith("_") and not hasattr(modules_shell, k):
                setattr(modules_shell, k, getattr(mod, k))
    cv_common = importlib.import_module("cflearn.modules.cv.common")
    for k in dir(cv_common):
        if not k.startswith("_") and not hasattr(modules_shell, k):
            setattr(modules_shell, k, getattr(cv_common, k))
    vit = importlib.import_module("cflearn.modules.cv.encoder.transformer")
    fcnn = importlib.import_module("cflearn.modules.ml.fcnn")
    # names the reference's package __init__ files would have re-exported (the shells skip them)
    setattr(sys.modules["cflearn.modules.cv.encoder"], "ViTEncoder", vit.ViTEncoder)

    _loaded = types.SimpleNamespace(
        common=common,
        core=core,
        toolkit=toolkit,
        Linear=core.Linear,
        Attention=core.Attention,
        FeedForward=core.FeedForward,
        MixingBlock=importlib.import_module("cflearn.modules.core.mixed_stacks.api").MixingBlock,
        MixedStackedEncoder=core.MixedStackedEncoder,
        NormFactory=core.NormFactory,
        Conv2d=core.Conv2d,
        sdp_attn=toolkit.sdp_attn,
        ViTEncoder=vit.ViTEncoder,
        FCNN=fcnn.FCNN,
        build_module=common.build_module,
        module_dict=common.module_dict,
    )
    return _loaded
