"""cftool.data_structures shells (the reference's API pool, api/common.py: inference serving, out of scope)."""
from typing import Any, Generic, TypeVar

T = TypeVar("T")


class IPoolItem:  # pragma: no cover - shell
    pass


class PoolItemContext:  # pragma: no cover - shell
    pass


class Pool(Generic[T]):  # pragma: no cover - shell
    def __init__(self, *a: Any, **k: Any) -> None:
        pass
