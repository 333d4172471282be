"""`cftool.array` names touched while importing the reference hot path.

Only `squeeze` and `to_torch` / `to_numpy` carry behaviour there
(reference call sites: cv/encoder/vanilla.py:156, high_level.py:95, toolkit.py:1182-1234).
"""
from typing import Any

import numpy as np
import torch


def is_string(arr: np.ndarray) -> bool:
    return np.issubdtype(arr.dtype, np.str_) or np.issubdtype(arr.dtype, np.object_)


def is_float(arr: np.ndarray) -> bool:
    return np.issubdtype(arr.dtype, np.floating)


def to_standard(arr: np.ndarray) -> np.ndarray:
    if np.issubdtype(arr.dtype, np.integer):
        return arr.astype(np.int64)
    if np.issubdtype(arr.dtype, np.floating):
        return arr.astype(np.float32)
    return arr


def to_torch(arr: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(to_standard(np.asarray(arr)))


def to_numpy(tensor: torch.Tensor) -> np.ndarray:
    return tensor.detach().cpu().numpy()


def to_device(batch: Any, device: Any, **kw: Any) -> Any:
    if isinstance(batch, dict):
        return {k: to_device(v, device, **kw) for k, v in batch.items()}
    if isinstance(batch, (list, tuple)):
        return type(batch)(to_device(v, device, **kw) for v in batch)
    if isinstance(batch, torch.Tensor):
        return batch.to(device, **kw)
    return batch


def squeeze(arr: Any) -> Any:
    """Drop every unit dimension, but keep the batch axis when batch size is 1."""
    n = arr.shape[0]
    arr = arr.squeeze()
    if n == 1:
        arr = arr[None, ...]
    return arr


def l2_normalize(arr: Any) -> Any:
    if isinstance(arr, np.ndarray):
        return arr / np.linalg.norm(arr, axis=-1, keepdims=True)
    return arr / arr.norm(dim=-1, keepdim=True)


def softmax(arr: Any) -> Any:
    if isinstance(arr, np.ndarray):
        e = np.exp(arr - arr.max(axis=1, keepdims=True))
        return e / e.sum(axis=1, keepdims=True)
    return torch.softmax(arr, dim=1)


def sigmoid(arr: Any) -> Any:
    if isinstance(arr, np.ndarray):
        return 1.0 / (1.0 + np.exp(-arr))
    return torch.sigmoid(arr)
