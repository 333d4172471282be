"""`cftool.pipeline` shells: imported by cflearn/schema.py at module import, unused on the hot path."""
from typing import Any, Dict, Generic, List, TypeVar

from .misc import ISerializable, WithRegister

T = TypeVar("T")
TConfig = TypeVar("TConfig")
TPipeline = TypeVar("TPipeline")


class IBlock(WithRegister["IBlock"]):
    d: Dict[str, Any] = {}
    previous: Dict[str, "IBlock"]

    def build(self, config: Any) -> None:  # pragma: no cover - shell
        pass

    @property
    def requirements(self) -> List[type]:
        return []


class IPipeline(ISerializable["IPipeline"]):
    d: Dict[str, Any] = {}
    blocks: List[Any]

    def __init__(self) -> None:
        self.blocks = []


def get_workspace(folder: str, *, force_new: bool = False) -> Any:  # pragma: no cover
    raise NotImplementedError("shell only")
