"""Import the reference's OWN package from /root/reference (read-only) and run it.

TEST INFRASTRUCTURE ONLY (build container only: /root/reference does not exist on the GPU box).
Used by `oracle/gen_golden.py`, by the `not gpu` tests that pin `oracle/*_oracle.py` against the reference, by the
tests that drive the reference's own trainer / step engine over this repo's modules, and by `bench.py`'s
`cpu_baseline` leg (kind "reference"); never by the product path.

Recipe (full-shim row): `import cflearn` — the real package, through its own `__init__.py` files —
works once three gaps are filled from here:
  * `cftool` (carefree-toolkit, un-vendored, pinned only as >= 0.3.12 by setup.py:45): ./cftool restates the names the
    reference imports, with behaviour where its call sites need it (registry, safe_execute, Serializer, IPipeline /
    IBlock, DataClassBase ...);
  * `torchvision` (not installed): ./stubs.py installs placeholder modules (datasets / PIL transforms are never used
    on the training path);
  * two version drifts: `importlib.metadata.version("carefree-learn")` (the tree is not pip-installed) and
    `ReduceLROnPlateau(verbose=...)` (keyword removed from torch 2.10) — patched in ./stubs.py, the reference itself
    is untouched.
"""
import importlib
import os
import sys
import types
from typing import Optional

REFERENCE_ROOT = os.environ.get("CFLEARN_REFERENCE_ROOT", "/root/reference")

_loaded: Optional[types.SimpleNamespace] = None


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "cflearn", "modules", "core"))


def import_cflearn() -> types.ModuleType:
    """`import cflearn` from the reference tree (the whole package: schema, trainer, pipeline, api ...)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    from . import stubs

    stubs.install_torchvision()
    stubs.patch_torch_compat()
    stubs.patch_package_version()
    if "cftool" not in sys.modules:
        importlib.import_module("cftool")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return importlib.import_module("cflearn")


def load_reference() -> types.SimpleNamespace:
    """Returns a namespace with the reference classes used by the oracle checks."""
    global _loaded
    if _loaded is not None:
        return _loaded
    cflearn = import_cflearn()
    common = importlib.import_module("cflearn.modules.common")
    core = importlib.import_module("cflearn.modules.core")
    toolkit = importlib.import_module("cflearn.toolkit")
    vit = importlib.import_module("cflearn.modules.cv.encoder.transformer")
    fcnn = importlib.import_module("cflearn.modules.ml.fcnn")
    _loaded = types.SimpleNamespace(
        cflearn=cflearn,
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
