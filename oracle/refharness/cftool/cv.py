"""cftool.cv stand-in: names the reference imports at module level (PIL / image IO helpers of its CV data blocks and
inference APIs).  None of them is on the training hot path; calling one raises."""
from typing import Any


def _unavailable(name: str) -> Any:
    def fn(*a: Any, **k: Any) -> Any:
        raise NotImplementedError(f"cftool.cv.{name} is not part of the oracle harness (image IO / PIL preprocessing)")

    fn.__name__ = name
    return fn


to_rgb = _unavailable("to_rgb")
to_uint8 = _unavailable("to_uint8")
to_alpha_channel = _unavailable("to_alpha_channel")
read_image = _unavailable("read_image")
save_images = _unavailable("save_images")
restrict_wh = _unavailable("restrict_wh")
get_suitable_size = _unavailable("get_suitable_size")


class ReadImageResponse:  # pragma: no cover - shell
    pass


class ImageProcessor:  # pragma: no cover - shell
    pass


class ImageBox:  # pragma: no cover - shell
    pass
