"""`cftool.misc` names touched while importing the reference hot path.

Six of them carry semantics there: `safe_execute`, `register_core`,
`shallow_copy_dict`, `update_dict`, `WithRegister`, and (in `array`) `squeeze`.  The rest are
import-time shells.
"""
import inspect
import json
import os
from dataclasses import asdict, dataclass, fields
from typing import Any, Callable, Dict, Generic, List, Optional, Type, TypeVar

T = TypeVar("T")


# -- printing ---------------------------------------------------------------------------------


def print_info(msg: str) -> None:
    print(f"> [ info ] {msg}")


def print_warning(msg: str) -> None:
    print(f"> [warning] {msg}")


def print_error(msg: str) -> None:
    print(f"> [ error ] {msg}")


def truncate_string_to_length(string: str, length: int) -> str:
    if len(string) <= length:
        return string
    half = (length - 5) // 2
    return f"{string[:half]} ... {string[-half:]}"


# -- dict helpers -----------------------------------------------------------------------------


def shallow_copy_dict(d: Any) -> Any:
    """Recursively rebuild dict / list containers; leaves are shared."""

    def _copy(v: Any) -> Any:
        if isinstance(v, dict):
            return {k: _copy(vv) for k, vv in v.items()}
        if isinstance(v, list):
            return [_copy(vv) for vv in v]
        return v

    return _copy(d)


def update_dict(src_dict: dict, tgt_dict: dict) -> dict:
    """Recursively merge `src_dict` INTO `tgt_dict` (src wins) and return `tgt_dict`."""
    for k, v in src_dict.items():
        tgt_v = tgt_dict.get(k)
        if isinstance(v, dict) and isinstance(tgt_v, dict):
            update_dict(v, tgt_v)
        else:
            tgt_dict[k] = v
    return tgt_dict


def prod(iterable: Any) -> Any:
    out = 1
    for x in iterable:
        out = out * x
    return out


# -- signature-aware call ---------------------------------------------------------------------


def get_arguments(*a: Any, **k: Any) -> Dict[str, Any]:  # pragma: no cover - shell
    raise NotImplementedError("shell only")


def check_requires(fn: Any, name: str, strict: bool = True) -> bool:
    if isinstance(fn, type):
        fn = fn.__init__  # type: ignore
    sig = inspect.signature(fn)
    for p in sig.parameters.values():
        if not strict and p.kind is inspect.Parameter.VAR_KEYWORD:
            return True
        if p.name == name:
            return True
    return False


def safe_execute(fn: Callable, kw: Dict[str, Any], *, strict: bool = False) -> Any:
    """Call `fn` with the subset of `kw` its signature accepts (everything if it has **kwargs)."""
    target = fn.__init__ if isinstance(fn, type) else fn  # type: ignore
    sig = inspect.signature(target)
    params = sig.parameters
    if any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values()):
        return fn(**kw)
    accepted = {k: v for k, v in kw.items() if k in params}
    return fn(**accepted)


# -- registries -------------------------------------------------------------------------------


def register_core(
    name: str,
    global_dict: Dict[str, Any],
    *,
    allow_duplicate: bool = False,
    before_register: Optional[Callable] = None,
    after_register: Optional[Callable] = None,
) -> Callable:
    def _register(cls: Any) -> Any:
        if before_register is not None:
            before_register(cls)
        if name in global_dict and not allow_duplicate:
            print_warning(f"'{name}' has already been registered, it will be skipped")
            return cls
        global_dict[name] = cls
        if after_register is not None:
            after_register(cls)
        return cls

    return _register


class WithRegister(Generic[T]):
    d: Dict[str, Any]
    __identifier__: str

    @classmethod
    def get(cls, name: str) -> Any:
        return cls.d[name]

    @classmethod
    def has(cls, name: str) -> bool:
        return name in cls.d

    @classmethod
    def make(cls, name: str, config: Dict[str, Any], *, ensure_safe: bool = False) -> Any:
        base = cls.get(name)
        if not ensure_safe:
            return base(**config)
        return safe_execute(base, config)

    @classmethod
    def make_multiple(cls, names: Any, configs: Any = None, *, ensure_safe: bool = False) -> Any:
        if configs is None:
            configs = {}
        if isinstance(names, str):
            return cls.make(names, configs, ensure_safe=ensure_safe)
        return [
            cls.make(n, shallow_copy_dict(configs.get(n, {})), ensure_safe=ensure_safe)
            for n in names
        ]

    @classmethod
    def register(cls, name: str, *, allow_duplicate: bool = False) -> Callable:
        def before(cls_: Type) -> None:
            cls_.__identifier__ = name

        return register_core(name, cls.d, allow_duplicate=allow_duplicate, before_register=before)

    @classmethod
    def remove(cls, name: str) -> Any:
        return cls.d.pop(name, None)

    @classmethod
    def check_subclass(cls, name: str) -> bool:
        return issubclass(cls.d[name], cls)


# -- dataclass / serialisation shells -----------------------------------------------------------


class DataClassBase:
    @property
    def field_names(self) -> List[str]:
        return [f.name for f in fields(self)]  # type: ignore

    def asdict(self) -> Dict[str, Any]:
        return asdict(self)  # type: ignore

    def copy(self) -> Any:
        return type(self)(**shallow_copy_dict(self.asdict()))

    def update_with(self, other: Any) -> Any:
        d = update_dict(other.asdict(), self.asdict())
        return type(self)(**d)


class ISerializable(WithRegister, Generic[T]):
    d: Dict[str, Any] = {}

    def to_info(self) -> Dict[str, Any]:  # pragma: no cover - shell
        return {}

    def from_info(self, info: Dict[str, Any]) -> None:  # pragma: no cover - shell
        pass


class PureFromInfoMixin:
    def from_info(self, info: Dict[str, Any]) -> None:
        for k, v in info.items():
            setattr(self, k, v)


class ISerializableArrays(ISerializable, Generic[T]):
    pass


class ISerializableDataClass(ISerializable, DataClassBase, Generic[T]):
    """Here the registry is a *classmethod* `d()` (reference: schema.py:497-499,1912-1914)."""

    @classmethod
    def get(cls, name: str) -> Any:
        return cls.d()[name]  # type: ignore

    @classmethod
    def has(cls, name: str) -> bool:
        return name in cls.d()  # type: ignore

    @classmethod
    def register(cls, name: str, *, allow_duplicate: bool = False) -> Callable:
        def before(cls_: Type) -> None:
            cls_.__identifier__ = name

        return register_core(
            name, cls.d(), allow_duplicate=allow_duplicate, before_register=before  # type: ignore
        )


class Serializer:  # pragma: no cover - shell
    pass


class OPTBase:
    def __init__(self) -> None:
        self._opt = dict(self.defaults)
        self.update_from_env()

    @property
    def env_key(self) -> str:
        raise NotImplementedError

    @property
    def defaults(self) -> Dict[str, Any]:
        raise NotImplementedError

    def __getattr__(self, name: str) -> Any:
        opt = self.__dict__.get("_opt", {})
        if name in opt:
            return opt[name]
        raise AttributeError(name)

    def update_from_env(self) -> None:
        raw = os.environ.get(self.env_key)
        if raw:
            self._opt.update(json.loads(raw))


class context_error_handler:
    def __enter__(self) -> Any:
        return self

    def _normal_exit(self, exc_type: Any, exc_val: Any, exc_tb: Any) -> None:
        pass

    def _exception_exit(self, exc_type: Any, exc_val: Any, exc_tb: Any) -> None:
        pass

    def __exit__(self, exc_type: Any, exc_val: Any, exc_tb: Any) -> None:
        if not exc_type:
            self._normal_exit(exc_type, exc_val, exc_tb)
        else:
            self._exception_exit(exc_type, exc_val, exc_tb)


class DownloadProgressBar:  # pragma: no cover - shell
    def __init__(self, *a: Any, **k: Any) -> None:
        pass
