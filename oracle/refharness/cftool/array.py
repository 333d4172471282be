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


# -- normalisation helpers of the tabular preprocessor (data/blocks/ml/preprocessor.py) -----------------------------
# each `xxx_normalize(arr, global_scale=False, return_stats=False)` / `xxx_normalize_from(arr, stats)` /
# `recover_xxx_normalize_from(arr, stats)` triple works column-wise on [N, F] arrays


def _stat(fn: Any, arr: np.ndarray, global_scale: bool) -> Any:
    return fn(arr).item() if global_scale else fn(arr, axis=0, keepdims=True)


def normalize(arr: np.ndarray, *, global_scale: bool = False, return_stats: bool = False) -> Any:
    mean, std = _stat(np.mean, arr, global_scale), _stat(np.std, arr, global_scale)
    std = np.maximum(std, 1.0e-8)
    out = (arr - mean) / std
    if not return_stats:
        return out
    stats = dict(mean=mean if global_scale else mean.tolist(), std=std.item() if global_scale else std.tolist())
    return out, stats


def normalize_from(arr: np.ndarray, stats: Any) -> np.ndarray:
    return (arr - np.array(stats["mean"], np.float32)) / np.array(stats["std"], np.float32)


def recover_normalize_from(arr: np.ndarray, stats: Any) -> np.ndarray:
    return arr * np.array(stats["std"], np.float32) + np.array(stats["mean"], np.float32)


def min_max_normalize(arr: np.ndarray, *, global_scale: bool = False, return_stats: bool = False) -> Any:
    lo, hi = _stat(np.min, arr, global_scale), _stat(np.max, arr, global_scale)
    diff = np.maximum(hi - lo, 1.0e-8)
    out = (arr - lo) / diff
    if not return_stats:
        return out
    stats = dict(min=lo if global_scale else lo.tolist(), diff=diff.item() if global_scale else diff.tolist())
    return out, stats


def min_max_normalize_from(arr: np.ndarray, stats: Any) -> np.ndarray:
    return (arr - np.array(stats["min"], np.float32)) / np.array(stats["diff"], np.float32)


def recover_min_max_normalize_from(arr: np.ndarray, stats: Any) -> np.ndarray:
    return arr * np.array(stats["diff"], np.float32) + np.array(stats["min"], np.float32)


def quantile_normalize(arr: np.ndarray, *, q: float = 0.01, global_scale: bool = False, return_stats: bool = False) -> Any:
    kw: Any = {} if global_scale else dict(axis=0, keepdims=True)
    med, lo, hi = np.median(arr, **kw), np.quantile(arr, q, **kw), np.quantile(arr, 1.0 - q, **kw)
    diff = np.maximum(hi - lo, 1.0e-8)
    out = (arr - med) / diff
    if not return_stats:
        return out
    stats = dict(median=np.asarray(med).tolist(), diff=np.asarray(diff).tolist())
    return out, stats


def quantile_normalize_from(arr: np.ndarray, stats: Any) -> np.ndarray:
    return (arr - np.array(stats["median"], np.float32)) / np.array(stats["diff"], np.float32)


def recover_quantile_normalize_from(arr: np.ndarray, stats: Any) -> np.ndarray:
    return arr * np.array(stats["diff"], np.float32) + np.array(stats["median"], np.float32)


# -- metric helpers (metrics.py) ----------------------------------------------------------------------------------


def get_full_logits(logits: np.ndarray) -> np.ndarray:
    """binary logits [N, 1] -> two-class logits [N, 2] (= [-x, x])"""
    if logits.shape[1] == 1:
        logits = np.concatenate([-logits, logits], axis=1)
    return logits


def get_label_predictions(logits: np.ndarray, threshold: float) -> np.ndarray:
    if logits.shape[1] == 1 or logits.shape[1] == 2:
        logits = get_full_logits(logits)
        if threshold == 0.5:
            return logits.argmax(1).reshape(-1, 1)
        probs = softmax(logits)
        return (probs[:, [1]] >= threshold).astype(np.int64)
    return logits.argmax(1).reshape(-1, 1)


def corr(predictions: Any, target: Any, weights: Any = None, *, get_diagonal: bool = False) -> Any:
    is_t = isinstance(predictions, torch.Tensor)
    p = predictions.detach().cpu().numpy() if is_t else predictions
    t = target.detach().cpu().numpy() if isinstance(target, torch.Tensor) else target
    p = p - p.mean(0, keepdims=True)
    t = t - t.mean(0, keepdims=True)
    mat = (p.T @ t) / (np.sqrt((p ** 2).sum(0))[:, None] * np.sqrt((t ** 2).sum(0))[None, :] + 1.0e-12)
    if get_diagonal:
        mat = np.diag(mat)
    return torch.from_numpy(np.asarray(mat)) if is_t else mat


def iou(logits: Any, labels: Any) -> Any:
    is_t = isinstance(logits, torch.Tensor)
    lg = logits.detach().cpu().numpy() if is_t else logits
    lb = labels.detach().cpu().numpy() if isinstance(labels, torch.Tensor) else labels
    n = lg.shape[0]
    if lg.shape[1] == 1:
        pred = (lg > 0).reshape(n, -1)
    else:
        pred = lg.argmax(1).reshape(n, -1) > 0
    tgt = lb.reshape(n, -1) > 0
    inter = (pred & tgt).sum(1)
    union = (pred | tgt).sum(1)
    out = inter / np.maximum(union, 1)
    return torch.from_numpy(out) if is_t else out


def get_unique_indices(arr: np.ndarray) -> Any:
    import types

    unique, inv, counts = np.unique(arr, return_inverse=True, return_counts=True)
    order = np.argsort(inv, kind="stable")
    split = np.split(order, np.cumsum(counts)[:-1])
    return types.SimpleNamespace(unique=unique, unique_cnt=counts, sorting_indices=order, split_arr=counts.cumsum()[:-1],
                                 split_indices=split)


tensor_dict_type = dict
