"""CPU restatement of the tabular encoder and of the dropout / DropPath arithmetic.  TEST INFRASTRUCTURE ONLY.

  * `encode` ........ `ml_encoder.Encoder.forward` + `CommonMLModel.encode` (reference modules/core/ml_encoder.py:171-209,
                      models/ml/common.py:67-87): categorical columns -> int64 indices with out-of-bound imputation
                      (value >= dim -> 0), one-hot blocks, embedding rows, merged as [numerical | one-hot | embedding]
  * `dropout` ....... nn.Dropout given its keep mask: `x * (mask / (1 - p))`, the noise formed in x's dtype
  * `drop_path` ..... customs.py:434-443 given its sample mask: `x.div(keep) * mask`

Pinned by oracle/gen_golden.py (gen_ml_encoder / gen_stochastic) against the reference's own modules and torch."""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor


def encode(x: Tensor, columns: List[int], dims: List[int], one_hot_columns: List[int], embedding_columns: List[int],
           tables: Dict[int, Tensor]) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor], Tensor]:
    """x f32 [B, F]; returns (indices int64 [B, K], one_hot, embedding, merged_all)."""
    cat = x[..., columns]
    dims_t = torch.tensor(dims, dtype=torch.float32)
    cat = torch.where(cat >= dims_t, torch.zeros_like(cat), cat)  # ml_encoder.py:176-183
    indices = cat.to(torch.long)
    dim_of = dict(zip(columns, dims))
    pos = {c: i for i, c in enumerate(columns)}
    one_hot = None
    if one_hot_columns:
        one_hot = torch.cat([F.one_hot(indices[..., pos[c]], dim_of[c]).to(torch.float32) for c in one_hot_columns], dim=-1)
    embedding = None
    if embedding_columns:
        embedding = torch.cat([F.embedding(indices[..., pos[c]], tables[c]) for c in embedding_columns], dim=-1)
    numerical = x[..., [c for c in range(x.shape[-1]) if c not in columns]]
    parts = [numerical] + [p for p in (one_hot, embedding) if p is not None]
    return indices, one_hot, embedding, torch.cat(parts, dim=-1)


def dropout(x: Tensor, mask: Tensor, p: float) -> Tensor:
    """torch's CPU dropout arithmetic for a given keep mask: `input * noise` with `noise = mask.div(1 - p)` formed IN
    THE INPUT'S DTYPE (ATen Dropout.cpp `_dropout_impl`): for bf16 tensors the scale 1 / (1 - p) is itself rounded to
    bf16 before the multiply."""
    return x * (mask.to(x.dtype) / (1.0 - p))


def drop_path(x: Tensor, mask: Tensor, keep_prob: float) -> Tensor:
    """customs.py:442: `net.div(keep_prob) * random_tensor` (two roundings in x's dtype)"""
    shape = (x.shape[0],) + (1,) * (x.dim() - 1)
    return x.div(keep_prob) * mask.to(x.dtype).view(shape)
