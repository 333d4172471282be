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


def get_arguments(*, num_back: int = 0, pop_class_attributes: bool = True) -> Dict[str, Any]:
    """The caller's (num_back frames up) local arguments, minus `self` / `__class__`."""
    frame = inspect.currentframe().f_back  # type: ignore
    for _ in range(num_back):
        frame = frame.f_back  # type: ignore
    args = dict(inspect.getargvalues(frame).locals)  # type: ignore
    if pop_class_attributes:
        args.pop("self", None)
        args.pop("__class__", None)
    return args


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


# -- dataclass / serialisation -------------------------------------------------------------------
# Restated from the reference's call sites (schema.py:294-330,493-600,636-661,1375-1390,1866-1949; pipeline/api.py:
# 275-311,389,543-569; pipeline/blocks/basic.py:116-181,748-752).  The on-disk layout only has to be self-consistent
# (SURVEY §8c): id.txt (registered name), info.json (`to_info()`), npd/ (one .npy per array).


class DataClassBase:
    @property
    def field_names(self) -> List[str]:
        return [f.name for f in fields(self)]  # type: ignore

    @property
    def attributes(self) -> List[Any]:
        return [getattr(self, name) for name in self.field_names]

    def asdict(self) -> Dict[str, Any]:
        return asdict(self)  # type: ignore

    def copy(self) -> Any:
        return type(self)(**shallow_copy_dict(self.asdict()))

    def update_with(self, other: Any) -> Any:
        d = update_dict(other.asdict(), self.asdict())
        return type(self)(**d)

    def as_tuple(self) -> Any:
        return tuple(self.attributes)

    @classmethod
    def construct(cls, d: Dict[str, Any]) -> Any:
        return safe_execute(cls, d)


@dataclass
class JsonPack(DataClassBase):
    type: str
    info: Dict[str, Any]


class ISerializable(WithRegister, Generic[T]):
    d: Dict[str, Any] = {}

    def to_info(self) -> Dict[str, Any]:
        return {}

    def from_info(self, info: Dict[str, Any]) -> None:
        pass

    def to_pack(self) -> JsonPack:
        return JsonPack(self.__identifier__, self.to_info())

    @classmethod
    def from_pack(cls, pack: Dict[str, Any]) -> Any:
        obj = cls.get(pack["type"])()
        obj.from_info(pack["info"])
        return obj

    def to_json(self) -> str:
        return json.dumps(self.to_pack().asdict())

    @classmethod
    def from_json(cls, json_string: str) -> Any:
        return cls.from_pack(json.loads(json_string))

    def copy(self) -> Any:
        copied = self.__class__()
        copied.from_info(shallow_copy_dict(self.to_info()))
        return copied


class PureFromInfoMixin:
    def from_info(self, info: Dict[str, Any]) -> None:
        for k, v in info.items():
            setattr(self, k, v)


class ISerializableArrays(ISerializable, Generic[T]):
    def to_npd(self) -> Dict[str, Any]:
        return {}

    def from_npd(self, npd: Dict[str, Any]) -> None:
        pass

    def copy(self) -> Any:
        copied = super().copy()
        copied.from_npd(shallow_copy_dict(self.to_npd()))
        return copied


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

    def to_info(self) -> Dict[str, Any]:
        return self.asdict()

    def from_info(self, info: Dict[str, Any]) -> None:
        for k, v in info.items():
            setattr(self, k, v)

    def copy(self) -> Any:
        return DataClassBase.copy(self)


class Serializer:
    id_file = "id.txt"
    info_file = "info.json"
    npd_folder = "npd"

    @classmethod
    def save_info(cls, folder: str, *, info: Optional[Dict[str, Any]] = None, serializable: Any = None) -> None:
        os.makedirs(folder, exist_ok=True)
        if info is None and serializable is None:
            raise ValueError("either `info` or `serializable` should be provided")
        if info is None:
            info = serializable.to_info()
        with open(os.path.join(folder, cls.info_file), "w") as f:
            json.dump(info, f)

    @classmethod
    def try_load_info(cls, folder: str, *, strict: bool = False) -> Optional[Dict[str, Any]]:
        path = os.path.join(folder, cls.info_file)
        if not os.path.isfile(path):
            if strict:
                raise ValueError(f"'{path}' does not exist")
            return None
        with open(path, "r") as f:
            return json.load(f)

    @classmethod
    def load_info(cls, folder: str) -> Dict[str, Any]:
        return cls.try_load_info(folder, strict=True)  # type: ignore

    @classmethod
    def save_npd(cls, folder: str, *, npd: Optional[Dict[str, Any]] = None, serializable: Any = None) -> None:
        import numpy as np

        os.makedirs(folder, exist_ok=True)
        if npd is None:
            npd = serializable.to_npd()
        npd_folder = os.path.join(folder, cls.npd_folder)
        os.makedirs(npd_folder, exist_ok=True)
        for k, v in npd.items():
            np.save(os.path.join(npd_folder, f"{k}.npy"), v)

    @classmethod
    def load_npd(cls, folder: str) -> Dict[str, Any]:
        import numpy as np

        npd_folder = os.path.join(folder, cls.npd_folder)
        if not os.path.isdir(npd_folder):
            return {}
        return {os.path.splitext(f)[0]: np.load(os.path.join(npd_folder, f), allow_pickle=True)
                for f in sorted(os.listdir(npd_folder)) if f.endswith(".npy")}

    @classmethod
    def save(cls, folder: str, serializable: Any, *, save_npd: bool = True) -> None:
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, cls.id_file), "w") as f:
            f.write(serializable.__identifier__)
        cls.save_info(folder, serializable=serializable)
        if save_npd and isinstance(serializable, ISerializableArrays):
            cls.save_npd(folder, serializable=serializable)

    @classmethod
    def load_empty(cls, folder: str, base: Any, *, swap_id: Optional[str] = None) -> Any:
        if swap_id is not None:
            s_id = swap_id
        else:
            with open(os.path.join(folder, cls.id_file), "r") as f:
                s_id = f.read().strip()
        return base.get(s_id)()

    @classmethod
    def load(cls, folder: str, base: Any, *, swap_id: Optional[str] = None,
             swap_info: Optional[Dict[str, Any]] = None, load_npd: bool = True) -> Any:
        serializable = cls.load_empty(folder, base, swap_id=swap_id)
        serializable.from_info(swap_info if swap_info is not None else cls.load_info(folder))
        if load_npd and isinstance(serializable, ISerializableArrays):
            serializable.from_npd(cls.load_npd(folder))
        return serializable


class Saving:
    @staticmethod
    def compress(abs_folder: str, remove_original: bool = True) -> None:
        import shutil

        shutil.make_archive(abs_folder, "zip", root_dir=os.path.dirname(abs_folder), base_dir=os.path.basename(abs_folder))
        if remove_original:
            shutil.rmtree(abs_folder)


class Incrementer:
    """Running mean / std over a sliding window (monitors.py:51-57,91-103)."""

    def __init__(self, window_size: Optional[int] = None):
        if window_size is not None and (not isinstance(window_size, int) or window_size < 2):
            raise ValueError("window_size should be an integer >= 2")
        self.window_size = window_size
        self._values: List[float] = []

    @property
    def is_full(self) -> bool:
        return self.window_size is not None and len(self._values) >= self.window_size

    def update(self, new_value: float) -> None:
        self._values.append(float(new_value))
        if self.window_size is not None and len(self._values) > self.window_size:
            self._values.pop(0)

    @property
    def mean(self) -> float:
        return sum(self._values) / max(len(self._values), 1)

    @property
    def std(self) -> float:
        n = len(self._values)
        if n == 0:
            return 0.0
        m = self.mean
        return (max(sum((v - m) ** 2 for v in self._values) / n, 0.0)) ** 0.5


class lock_manager:
    """Directory lock used by the mlflow callback only (callbacks/general.py:137): a plain context here."""

    def __init__(self, workspace: str, stuffs: List[str], **kwargs: Any):
        self._workspace, self._stuffs = workspace, stuffs

    def __enter__(self) -> "lock_manager":
        os.makedirs(self._workspace, exist_ok=True)
        return self

    def __exit__(self, *exc: Any) -> None:
        return None


def random_hash() -> str:
    import uuid

    return uuid.uuid4().hex


def hash_dict(d: Dict[str, Any]) -> str:
    import hashlib

    return hashlib.md5(json.dumps(d, sort_keys=True, default=str).encode()).hexdigest()


def sort_dict_by_value(d: Dict[Any, Any], *, reverse: bool = False) -> Dict[Any, Any]:
    return dict(sorted(d.items(), key=lambda kv: kv[1], reverse=reverse))


def fix_float_to_length(num: float, length: int) -> str:
    import math

    if length <= 0:
        return ""
    if isinstance(num, float) and math.isnan(num):
        return "nan".ljust(length)
    s = f"{num:.{max(length, 1)}f}" if isinstance(num, float) else str(num)
    if "." in s and len(s) > length:
        s = s[:length]
    return s.ljust(length, "0" if "." in s else " ")


def is_numeric(s: Any) -> bool:
    try:
        float(s)
        return True
    except (TypeError, ValueError):
        return False


def walk(root: str, hierarchy_callback: Callable, filter_extensions: Any = None) -> None:
    for folder, _, files in os.walk(root):
        for file in files:
            if filter_extensions is not None and os.path.splitext(file)[1] not in filter_extensions:
                continue
            hierarchy_callback(folder.split(os.path.sep), os.path.join(folder, file))


def timestamp(simplify: bool = False, ensure_different: bool = False) -> str:
    import datetime

    now = datetime.datetime.now()
    if simplify:
        return now.strftime("%Y-%m-%d")
    if ensure_different:
        return now.strftime("%Y-%m-%d_%H-%M-%S-%f")
    return now.strftime("%Y-%m-%d_%H-%M-%S")


def prepare_workspace_from(workspace: str, *, timeout: Any = None, make: bool = True) -> str:
    """`<workspace>/<timestamp>` (a fresh sub-folder per run; pipeline/api.py:275, api/api.py:543)."""
    current = timestamp(ensure_different=True)
    ws = os.path.join(workspace, current)
    if make:
        os.makedirs(ws, exist_ok=True)
    return ws


def get_latest_workspace(root: str) -> Optional[str]:
    if not os.path.isdir(root):
        return None
    subs = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
    return None if not subs else os.path.join(root, subs[-1])


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
