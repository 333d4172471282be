"""Type aliases the reference imports from `cftool.types` (annotations only)."""
from typing import Any, Dict, Union

import numpy as np

try:  # torch is present in the harness; keep the import soft for tooling
    import torch

    arr_type = Union[np.ndarray, "torch.Tensor"]
    tensor_dict_type = Dict[str, Any]
except Exception:  # pragma: no cover
    arr_type = Any
    tensor_dict_type = Dict[str, Any]

np_dict_type = Dict[str, Any]
general_config_type = Any
configs_type = Any

TNumberPair = Any
