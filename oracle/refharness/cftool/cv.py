"""cftool.cv stand-in: only the name the reference CLIP module imports (used by its PIL transform, never called
by the oracle)."""
from typing import Any


def to_rgb(image: Any, color: Any = None) -> Any:
    raise NotImplementedError("cftool.cv.to_rgb is not part of the oracle harness (PIL preprocessing)")
