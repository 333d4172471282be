"""cftool.dist.Parallel shell (process task farm of cflearn/dist and the image-folder preprocessing; SURVEY F4:
not a communication backend, out of scope)."""
from typing import Any


class Parallel:  # pragma: no cover - shell
    def __init__(self, *a: Any, **k: Any) -> None:
        pass

    def __call__(self, *a: Any, **k: Any) -> Any:
        raise NotImplementedError("cftool.dist.Parallel is not part of the oracle harness")
