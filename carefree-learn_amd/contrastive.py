"""CLIP training step pieces that the reference does NOT have (SURVEY §2 row "multimodal/clip.py": inference only, no
contrastive loss, no all-gather; BASELINE config 5 asks for both): **new design, parity unpinned** — the towers and
`logit_scale.exp() * I @ T^T` are pinned (tests/golden/clip_small.pt), the loss below is the standard symmetric
InfoNCE of the CLIP paper and is tested against a plain fp32 PyTorch statement of the same formula.

    all_I, all_T = gather over ranks (autograd-aware)          # [W*B, D]
    L = ( CE(s * I_local @ all_T^T, y) + CE(s * T_local @ all_I^T, y) ) / 2,   y_i = rank * B + i,  s = exp(logit_scale)

Every rank computes the rows of its OWN samples against the embeddings of ALL ranks ("local loss"); the backward of the
gather sums, over the ranks, the gradient each of them holds for this rank's embeddings (reduce-scatter on RCCL,
all-reduce + slice on gloo).  With the 1/W of the data-parallel gradient average this is the gradient of the global
mean loss.  Arithmetic: fp32 similarity GEMMs (`cfhip_sgemm_f32`; the L2-normalised features are not rounded to bf16),
the softmax-CE kernel of the classifier path, `cfhip_dot_f32` for d logit_scale.
"""
from typing import Any, Optional

import torch
import torch.distributed as dist
from torch import Tensor
from torch.autograd import Function

from . import ops

f32 = torch.float32


class GatherRowsWithGradFn(Function):
    """[B, D] on every rank -> [W*B, D] (rank-major); backward: this rank's slice of the SUM over ranks of the
    incoming gradients."""

    @staticmethod
    def forward(ctx: Any, x: Tensor, group: Any) -> Tensor:
        ctx.group = group
        world = dist.get_world_size(group)
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        if x.is_cuda:
            dist.all_gather_into_tensor(out, x, group=group)  # RCCL ncclAllGather
        else:
            dist.all_gather(list(out.chunk(world, dim=0)), x, group=group)  # gloo (CPU tests)
        ctx.rows = x.shape[0]
        return out

    @staticmethod
    def backward(ctx: Any, g: Tensor):  # type: ignore
        group, rows = ctx.group, ctx.rows
        rank = dist.get_rank(group)
        g = g.contiguous()
        if g.is_cuda:
            mine = torch.empty((rows,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(mine, g, op=dist.ReduceOp.SUM, group=group)  # RCCL ncclReduceScatter
            return mine, None
        g = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)  # gloo has no reduce-scatter
        return g[rank * rows:(rank + 1) * rows].clone(), None


def gather_rows_with_grad(x: Tensor, group: Any = None) -> Tensor:
    return GatherRowsWithGradFn.apply(x, group)


class ClipLossFn(Function):
    """Local rows of the symmetric InfoNCE loss; returns the mean over the local batch (f32 [1] on the device)."""

    @staticmethod
    def forward(ctx: Any, img: Tensor, txt: Tensor, all_img: Tensor, all_txt: Tensor, logit_scale: Tensor,
                offset: int) -> Tensor:
        img, txt, all_img, all_txt = (t.float().contiguous() for t in (img, txt, all_img, all_txt))
        b = img.shape[0]
        s = logit_scale.detach().float().exp().reshape(1)  # device scalar: no host synchronisation
        li = ops.sgemm_f32(img, all_txt, alpha_dev=s)  # logits_per_image  [B, W*B]
        lt = ops.sgemm_f32(txt, all_img, alpha_dev=s)  # logits_per_text   [B, W*B]
        labels = torch.arange(offset, offset + b, dtype=torch.int64, device=img.device)
        sum_i, dli = ops.softmax_xent(li, labels, 0.5 / b)
        sum_t, dlt = ops.softmax_xent(lt, labels, 0.5 / b)
        ctx.save_for_backward(img, txt, all_img, all_txt, li, lt, dli, dlt, s)
        ctx.ls_shape = tuple(logit_scale.shape)
        return (sum_i + sum_t) * (0.5 / b)

    @staticmethod
    def backward(ctx: Any, g: Tensor):  # type: ignore
        img, txt, all_img, all_txt, li, lt, dli, dlt, s = ctx.saved_tensors
        a = (s * g.float().reshape(1)).contiguous()  # s * upstream gradient, still on the device
        d_img = ops.sgemm_f32(dli, all_txt, b_trans=True, alpha_dev=a)                    # dLi @ all_T     [B, D]
        d_all_txt = ops.sgemm_f32(dli, img, a_trans=True, b_trans=True, alpha_dev=a)      # dLi^T @ I       [W*B, D]
        d_txt = ops.sgemm_f32(dlt, all_img, b_trans=True, alpha_dev=a)
        d_all_img = ops.sgemm_f32(dlt, txt, a_trans=True, b_trans=True, alpha_dev=a)
        # logits = s * P  =>  dL/d(log s) = sum(dlogits * logits)
        d_ls = (ops.dot_f32(dli, li) + ops.dot_f32(dlt, lt)) * g.float().reshape(1)
        # (logit_scale is a 0-d parameter in the reference, multimodal/schema.py:15)
        return d_img, d_txt, d_all_img, d_all_txt, d_ls.reshape(ctx.ls_shape), None


class SimilarityFn(Function):
    """`logit_scale.exp() * image_features @ text_features.t()` (reference multimodal/schema.py:25-30) in fp32."""

    @staticmethod
    def forward(ctx: Any, img: Tensor, txt: Tensor, logit_scale: Tensor) -> Tensor:
        img, txt = img.float().contiguous(), txt.float().contiguous()
        s = logit_scale.detach().float().exp().reshape(1)
        logits = ops.sgemm_f32(img, txt, alpha_dev=s)
        ctx.save_for_backward(img, txt, logits, s)
        ctx.ls_shape = tuple(logit_scale.shape)
        return logits

    @staticmethod
    def backward(ctx: Any, g: Tensor):  # type: ignore
        img, txt, logits, s = ctx.saved_tensors
        g = g.float().contiguous()
        d_img = ops.sgemm_f32(g, txt, b_trans=True, alpha_dev=s)
        d_txt = ops.sgemm_f32(g, img, a_trans=True, b_trans=True, alpha_dev=s)
        return d_img, d_txt, ops.dot_f32(g, logits).reshape(ctx.ls_shape)


def similarity_logits(image_features: Tensor, text_features: Tensor, logit_scale: Tensor) -> Tensor:
    return SimilarityFn.apply(image_features, text_features, logit_scale)


def clip_contrastive_loss(image_features: Tensor, text_features: Tensor, logit_scale: Tensor,
                          group: Optional[Any] = None) -> Tensor:
    """Mean symmetric InfoNCE over the LOCAL batch against the embeddings of every rank of `group` (None + an
    initialised default group: the world; not initialised: single process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        all_img = gather_rows_with_grad(image_features, group)
        all_txt = gather_rows_with_grad(text_features, group)
        offset = dist.get_rank(group) * image_features.shape[0]
    else:
        all_img, all_txt, offset = image_features, text_features, 0
    return ClipLossFn.apply(image_features, text_features, all_img, all_txt, logit_scale, offset)
