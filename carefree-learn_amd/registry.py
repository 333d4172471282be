"""Module registry with the reference's surface (cflearn/modules/common.py:30-83).

`register_module(name)` / `build_module(name, config=...)` / `PrefixModules(prefix)` behave as in
the reference: configs are dicts (or a json path), keyword arguments the constructor does not
accept are silently dropped (`safe_execute`, what `cftool.misc.safe_execute` does at
modules/common.py:52-53), missing required ones raise TypeError.  Unlike the reference's default,
re-registering a name REPLACES the entry (that is how the HIP modules take over reference names
when both packages are imported — see INTEGRATION.md).
"""
import inspect
import json
from typing import Any, Callable, Dict, List, Optional, Type, Union

from torch.nn import Module

module_dict: Dict[str, Type[Module]] = {}


def shallow_copy_dict(d: Any) -> Any:
    if isinstance(d, dict):
        return {k: shallow_copy_dict(v) for k, v in d.items()}
    if isinstance(d, list):
        return [shallow_copy_dict(v) for v in d]
    return d


def update_dict(src: Dict[str, Any], tgt: Dict[str, Any]) -> Dict[str, Any]:
    """merge `src` into `tgt` recursively (src wins); returns `tgt`."""
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(tgt.get(k), dict):
            update_dict(v, tgt[k])
        else:
            tgt[k] = v
    return tgt


def safe_execute(fn: Callable, kw: Dict[str, Any]) -> Any:
    target = fn.__init__ if isinstance(fn, type) else fn
    params = inspect.signature(target).parameters
    if any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values()):
        return fn(**kw)
    return fn(**{k: v for k, v in kw.items() if k in params})


def register_module(name: str, **_: Any) -> Callable[[Type[Module]], Type[Module]]:
    def deco(cls: Type[Module]) -> Type[Module]:
        module_dict[name] = cls
        return cls

    return deco


def build_module(name: str, *, config: Optional[Union[str, Dict[str, Any]]] = None, **kwargs: Any) -> Module:
    if config is None:
        kw = shallow_copy_dict(kwargs)
    else:
        if not isinstance(config, dict):
            with open(config, "r") as f:
                config = json.load(f)
        kw = shallow_copy_dict(config)
        update_dict(shallow_copy_dict(kwargs), kw)
    if name not in module_dict:
        raise KeyError(f"module '{name}' is not registered (known: {sorted(module_dict)})")
    return safe_execute(module_dict[name], kw)


class PrefixModules:
    """Namespaced view of the registry: `PrefixModules("attention").build("basic", ...)`."""

    def __init__(self, prefix: str) -> None:
        self._prefix = prefix

    def prefix(self, name: str) -> str:
        return f"{self._prefix}.{name}"

    @property
    def all(self) -> List[str]:
        return sorted(k for k in module_dict if k.startswith(self._prefix))

    def has(self, name: str) -> bool:
        return self.prefix(name) in module_dict

    def get(self, name: str) -> Optional[Type[Module]]:
        return module_dict.get(self.prefix(name))

    def register(self, name: str, **kwargs: Any) -> Callable[[Type[Module]], Type[Module]]:
        return register_module(self.prefix(name), **kwargs)

    def build(self, name: str, *, config: Any = None, **kwargs: Any) -> Module:
        return build_module(self.prefix(name), config=config, **kwargs)


attentions = PrefixModules("attention")
token_mixers = PrefixModules("token_mixer")
channel_mixers = PrefixModules("channel_mixer")
encoders = PrefixModules("encoders")


def override_reference_registry(reference_module_dict: Dict[str, Any], names: Optional[List[str]] = None) -> int:
    """INTEGRATION.md §2 as one call: put this package's classes into the REFERENCE's registry
    (`cflearn.modules.common.module_dict`) under the reference's own names, so that the reference's unchanged
    `build_module` (modules/common.py:37-53 — how `CommonDLModel.build`, `build_encoder`, `build_attention` ... create
    every module) returns the HIP-backed classes.  Direct dict assignment, because the reference's
    `register_module` warns and SKIPS an existing name (cftool `register_core`, allow_duplicate=False).
    `names`: restrict the override (default: every name this package registers).  Returns the number of entries set."""
    count = 0
    for name, cls in module_dict.items():
        if names is not None and name not in names:
            continue
        reference_module_dict[name] = cls
        count += 1
    return count
