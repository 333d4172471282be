"""`cftool.pipeline`: `IBlock` / `IPipeline` as the reference uses them (pipeline/common.py:19-129, schema.py:364-590,
pipeline/api.py:129-300,545-572).  A pipeline is an ordered list of registered blocks; `build(*blocks)` checks every
block's `requirements` against what was built before it, hands it `previous`, and calls
before_block_build -> block.build(config) -> after_block_build."""
import os
import shutil
import tempfile
from typing import Any, Dict, Generic, List, Optional, Type, TypeVar

from .misc import ISerializable, WithRegister, shallow_copy_dict

T = TypeVar("T")
TConfig = TypeVar("TConfig")
TPipeline = TypeVar("TPipeline")
TBlock = TypeVar("TBlock")


class IBlock(WithRegister["IBlock"]):
    d: Dict[str, Any] = {}
    previous: Dict[str, "IBlock"]

    def build(self, config: Any) -> None:
        pass

    @property
    def requirements(self) -> List[type]:
        return []

    def try_get_previous(self, block: Any) -> Any:
        if not isinstance(block, str):
            block = block.__identifier__
        return self.previous.get(block)

    def get_previous(self, block: Any) -> Any:
        b = self.try_get_previous(block)
        if b is None:
            raise ValueError(f"cannot find '{block}' in `previous`")
        return b


def _check_requirement(block: IBlock, previous: Dict[str, IBlock]) -> None:
    for requirement in block.requirements:
        if requirement.__identifier__ not in previous:
            raise ValueError(f"'{type(block).__name__}' requires '{requirement.__name__}', "
                             "but none is provided in the previous blocks")


class IPipeline(ISerializable["IPipeline"]):
    d: Dict[str, Any] = {}
    config: Any
    blocks: List[Any]

    def __init__(self) -> None:
        self.blocks = []

    @classmethod
    def init(cls, config: Any) -> Any:
        raise NotImplementedError

    @property
    def config_base(self) -> Any:
        raise NotImplementedError

    @property
    def block_base(self) -> Any:
        raise NotImplementedError

    def to_info(self) -> Dict[str, Any]:
        return dict(blocks=[b.__identifier__ for b in self.blocks], config=self.config.to_pack().asdict())

    def from_info(self, info: Dict[str, Any]) -> None:
        self.config = self.config_base.from_pack(info["config"])
        block_types = [self.block_base.get(b) for b in info["blocks"]]
        self.build(*[t() for t in block_types])
        self.after_load()

    # optional callbacks

    def before_block_build(self, block: Any) -> None:
        pass

    def after_block_build(self, block: Any) -> None:
        pass

    def after_load(self) -> None:
        pass

    # api

    @property
    def block_mappings(self) -> Dict[str, Any]:
        return {b.__identifier__: b for b in self.blocks}

    def try_get_block(self, block: Any) -> Any:
        if not isinstance(block, str):
            block = block.__identifier__
        return self.block_mappings.get(block)

    def get_block(self, block: Any) -> Any:
        b = self.try_get_block(block)
        if b is None:
            raise ValueError(f"cannot find '{block}' in `blocks`")
        return b

    def remove(self, *block_names: str) -> None:
        drop = set(block_names)
        self.blocks = [b for b in self.blocks if b.__identifier__ not in drop]

    def build(self, *blocks: Any) -> None:
        previous: Dict[str, Any] = self.block_mappings
        for block in blocks:
            _check_requirement(block, previous)
            block.previous = shallow_copy_dict(previous)
            self.before_block_build(block)
            block.build(self.config)
            self.after_block_build(block)
            previous[block.__identifier__] = block
            self.blocks.append(block)


class get_workspace:
    """A folder or its .zip as a readable directory (pipeline/api.py:543,638,658); `force_new` = work on a copy."""

    def __init__(self, folder: str, *, force_new: bool = False):
        self.folder, self.force_new = folder, force_new
        self._tmp: Optional[str] = None

    def __enter__(self) -> str:
        if os.path.isdir(self.folder):
            if not self.force_new:
                return self.folder
            self._tmp = tempfile.mkdtemp()
            dst = os.path.join(self._tmp, os.path.basename(os.path.normpath(self.folder)))
            shutil.copytree(self.folder, dst)
            return dst
        zip_path = self.folder if self.folder.endswith(".zip") else f"{self.folder}.zip"
        if not os.path.isfile(zip_path):
            raise ValueError(f"neither '{self.folder}' nor '{zip_path}' exists")
        self._tmp = tempfile.mkdtemp()
        shutil.unpack_archive(zip_path, self._tmp, "zip")
        subs = os.listdir(self._tmp)
        return os.path.join(self._tmp, subs[0]) if len(subs) == 1 else self._tmp

    def __exit__(self, *exc: Any) -> None:
        if self._tmp is not None:
            shutil.rmtree(self._tmp, ignore_errors=True)
