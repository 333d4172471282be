"""Import-time stand-ins for optional third-party packages the reference imports at module level but never needs
on the training hot path (torchvision: datasets / PIL transforms; the package is not installed and there is no
network).  TEST INFRASTRUCTURE ONLY.  Every name resolves to a placeholder class; USING one raises."""
import sys
import types
from typing import Any


class _Placeholder:
    def __init__(self, *a: Any, **k: Any) -> None:
        raise NotImplementedError(f"{type(self).__module__}.{type(self).__name__} is a placeholder of the oracle harness")


class _LazyModule(types.ModuleType):
    """Attribute access creates placeholder classes (CamelCase / lower names alike) or sub-modules on demand."""

    def __getattr__(self, name: str) -> Any:
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (_Placeholder,), {"__module__": self.__name__})
        setattr(self, name, cls)
        return cls


def install_torchvision() -> None:
    if "torchvision" in sys.modules:
        return
    names = ["torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.transforms.functional",
             "torchvision.utils", "torchvision.models", "torchvision.ops", "torchvision.io"]
    for n in names:
        m = _LazyModule(n)
        m.__path__ = []  # type: ignore
        sys.modules[n] = m
        if "." in n:
            parent, child = n.rsplit(".", 1)
            setattr(sys.modules[parent], child, m)
    sys.modules["torchvision.transforms"].InterpolationMode = types.SimpleNamespace(  # type: ignore
        BICUBIC="bicubic", BILINEAR="bilinear", NEAREST="nearest")
    sys.modules["torchvision"].__version__ = "0.0.0-stub"  # type: ignore


def patch_torch_compat() -> None:
    """The reference (2023) passes keywords that torch 2.10 has since removed; accept and drop them so its own code runs
    unmodified: `ReduceLROnPlateau(verbose=...)` (schedulers.py:140)."""
    import torch

    cls = torch.optim.lr_scheduler.ReduceLROnPlateau
    if getattr(cls, "_cfhip_patched", False):
        return
    orig = cls.__init__

    def init(self: Any, *a: Any, verbose: Any = None, **k: Any) -> None:
        orig(self, *a, **k)

    cls.__init__ = init  # type: ignore
    cls._cfhip_patched = True  # type: ignore


def patch_package_version() -> None:
    """`cflearn/__init__.py` ends with importlib.metadata.version("carefree-learn"); the tree is not pip-installed."""
    import importlib.metadata as md

    if getattr(md, "_cfhip_patched", False):
        return
    orig = md.version
    md.version = lambda name: "0.5.0" if name == "carefree-learn" else orig(name)  # type: ignore
    md._cfhip_patched = True  # type: ignore
