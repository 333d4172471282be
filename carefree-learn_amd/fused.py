"""Fused pre-norm transformer block: ONE autograd node per `MixingBlock`.

Forward (reference mixed_stacks/api.py:130-158 with attention token mixer + FeedForward channel
mixer, dropout = drop_path = 0):

    x1 = x  + out_linear(attn(split_heads(LN1(x) in_w^T + qkv_bias)))
    x2 = x1 + W2 gelu(W1 LN2(x1) + b1) + b2

8 launches forward: LN, GEMM(+bias), flash attention (reads the packed qkv in place, writes merged
heads), GEMM(+bias +residual), LN, GEMM(+bias +GELU, also stores the pre-activation),
GEMM(+bias +residual).  Backward: every dX GEMM reads W row-major through the transposing LDS read
(no W^T copies), every dW GEMM reads dY and X token-major the same way (no activation transposes),
GELU' is the epilogue of the FF2 dX GEMM, both residual-gradient adds are folded into the LayerNorm
backward kernels, and all parameter gradients are written directly into `param.grad`.
"""
import os
from typing import Any, Optional

import torch
from torch import Tensor
from torch.autograd import Function

from . import functional as _functional
from . import ops
from .functional import SideStream, bf16, f32, grad_buffer, shadow_bf16


# The dW GEMM also produces db = colsum(dy) (cfhip_gemm_bf16's `bias_grad`: the waves that own the first tile column
# re-read their A fragments and add them up).  Measured on ViT-B/16 batch 128 (tools/step_ab.py, interleaved rounds):
# 23.86 ms fused vs 23.98 ms with the stand-alone column-sum kernels on the second side stream; no bias gradients at
# all would be 23.63 ms, so this item is closed.
FUSE_BIAS_GRAD = True
_SKIP_BIAS_GRAD = False  # timing experiments only (tools/step_ab.py): leaves the bias gradients unwritten


def _dw_db(w: Tensor, b: Optional[Tensor], dy2: Tensor, x2: Tensor) -> None:
    """dW = dy^T x and db = colsum(dy) from ONE split-K GEMM launch (the bias gradient rides on a
    ones-operand MFMA inside the dW kernel), both written straight into `.grad`."""
    from .functional import grad_ready_callbacks

    n, k = w.shape[0], x2.shape[1]
    split = ops.pick_split_k(n, k, x2.shape[0])
    prms = [w] + ([b] if (b is not None and b.requires_grad) else [])
    for prm in prms:
        if prm.grad is None:
            prm.grad = grad_buffer(prm)
            prm._cfhip_fresh = True
    acc_w = not getattr(w, "_cfhip_fresh", False)
    kw = {}
    if len(prms) == 2:
        acc_b = not getattr(b, "_cfhip_fresh", False)
        if _SKIP_BIAS_GRAD:
            pass
        elif FUSE_BIAS_GRAD:
            kw = dict(bias_grad=b.grad.view(-1), bias_grad_accumulate=acc_b)
        else:  # second side stream: the column sum runs beside the dW GEMM
            SideStream.run(lambda: ops.colsum(dy2, out=b.grad.view(-1), accumulate=acc_b), (dy2,), lane=1)
    ops.gemm(dy2, x2, a_trans=True, b_trans=True, out=w.grad.view(n, k), accumulate=acc_w, split_k=split, **kw)
    for prm in prms:
        _functional.notify_grad_ready(prm)


# Round 3: the weight gradients of DW_GROUP_BLOCKS consecutive blocks (4 GEMMs each) go out as ONE grouped launch
# (ops.gemm_grouped_tn: 256 x 256 tiles of all problems share the chip, every tile runs its whole K = batch * tokens
# reduction — no split-K slabs, no reduce launches).  ViT-B/16: 108 tiles per block, two blocks = 216 tiles = one round on
# 256 CUs.  0 = the round-1/2 path (one split-K GEMM + reduce per weight gradient, issued beside its sibling dX GEMM).
DW_GROUP_BLOCKS = 2
# When a block has far fewer than 108 tiles (the CLIP text tower: 512 wide = 48 tiles per block) two blocks leave most of the
# chip idle: with DW_GROUP_TILES > 0 the queue is flushed by TILE count instead — as soon as another block like the last one
# would not fit into one round of 256-tile slots — and DW_GROUP_BLOCKS only says "grouping on" (ViT-B/16: still two blocks).
DW_GROUP_TILES = 256
# The LAST blocks of a backward pass (block indices below this) are flushed one by one: when the dX chain ends, what is left on
# the weight-gradient lane is then one block's launch (108 tiles) instead of two blocks' (216) — nothing else is left to
# share the chip with at that point, so the tail of the step is that launch's duration.
DW_TAIL_BLOCKS = 0  # measured (profiles/r04/dw_tail_ab.txt): 1 .. 3 cost 0.3 ms per step — whole-reduction tiles on 108 of 256 CUs; kept as a knob
DW_GROUP_ON_MAIN = False  # True: the grouped launch runs on the caller's stream (after the blocks' dX chain) instead of the side stream
_pending_dw: list = []
_slice_streams: list = []  # streams of the backward's batch slices beyond the caller's (what a dW launch has to wait for)


# Stand-alone Linear layers (functional.LinearFn: the UNet's ~600 small projections, CLIP projections, classifier heads, the
# patch embedding) queue their weight gradients here too: flushed when LINEAR_DW_TILES tiles are waiting, together with the next
# block-stack flush, or when the backward pass ends.  A flush of fewer than DW_MIN_TILES tiles falls back to one split-K GEMM
# per gradient (whole-reduction 256 x 256 tiles on a handful of CUs would take longer than 128 x 128 split-K tiles on all).
# Measured flat on the 64^2 x 8 UNet in rounds 3, 4 and 6 (profiles/r06/unet_variants_ab.txt; 55.7 vs 55.4 ms again at the end of round 6) and
# worth 4-5 ms of the 256^2 x 1 step (460.9 vs 464.8 ms with the level-0 layers alone, 459.6 vs 465.0 with every layer from 12 288 rows on:
# profiles/r06/unet256_linear_dw_grouped_ab.txt): there the token-level projections reduce 65 536 rows into 320 x 320 outputs, and a
# whole-reduction 256 x 256 tile streams its operands once where the split form writes and re-reads 57-114 slabs per gradient.  So: on for
# reductions of at least LINEAR_DW_MIN_ROWS rows.  LINEAR_DW_TILES = 0: LinearFn computes every weight gradient on the spot (rounds 1-2).
LINEAR_DW_TILES = 256
LINEAR_DW_MIN_ROWS = 49152
DW_MIN_TILES = 48
GROUP_MAX = 24  # problems per launch (gemm_grouped.hip: the by-value problem table)
_end_flush_queued = False


def _groupable(w: Tensor, dy2: Tensor, x2: Tensor) -> bool:
    n, k = dy2.shape[1], x2.shape[1]
    g = w.grad  # a user-replaced `.grad` that the grouped kernel could not write (strided / unaligned): the split-K path takes it
    if g is not None and (not g.is_contiguous() or g.data_ptr() % 16 != 0 or g.dtype != f32):
        return False
    return (dy2.is_cuda and w.numel() == n * k and w.is_contiguous() and n % 8 == 0 and k % 8 == 0 and dy2.stride(0) % 8 == 0
            and x2.stride(0) % 8 == 0 and x2.shape[0] * max(dy2.stride(0), x2.stride(0)) * 2 < 2 ** 31
            and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0)


# The last blocks of a backward pass (block index < DW_PEROP_TAIL) issue their weight gradients per operator (split-K GEMM +
# reduce on the weight-gradient lane, beside the dX chain) instead of queueing them: the grouped launch of the final blocks
# can only start when the chain has ended, and its ~0.9 ms are the tail of the step.  Measured (tools/step_variants.py
# "weight gradients per operator (DW_PEROP"): 1 block +0.29 ms, 2 blocks +0.41 ms — the grouped launch stays; kept as a knob.
DW_PEROP_TAIL = 0
_perop_now = False


def _queue_dw(w: Tensor, b: Optional[Tensor], dy2: Tensor, x2: Tensor) -> None:
    """dW = dy^T x (+ db): queued for the next grouped launch, or issued now on the side stream (round-1/2 path)."""
    if DW_GROUP_BLOCKS <= 0 or _perop_now or not _groupable(w, dy2, x2):
        SideStream.run(lambda: _dw_db(w, b, dy2, x2), (dy2, x2), wait=tuple(_slice_streams))
        return
    _pending_dw.append((w, b, dy2, x2))


def queue_linear_dw(w: Tensor, b: Optional[Tensor], dy2: Tensor, x2: Tensor) -> bool:
    """`functional.LinearFn.backward`: take over dW (+ db) of a stand-alone Linear whose gradients go straight into `.grad`.
    False = not taken (grouping off, shapes the grouped kernel does not accept, not inside a backward pass)."""
    global _end_flush_queued
    if LINEAR_DW_TILES <= 0 or DW_GROUP_BLOCKS <= 0 or not w.requires_grad or x2.shape[0] < LINEAR_DW_MIN_ROWS or not _groupable(w, dy2, x2):
        return False
    if any(not getattr(getattr(cb, "__self__", None), "accepts_deferred_gradients", False) for cb in _functional.grad_ready_callbacks):
        # A gradient reducer is listening (optim.StepInBackward only acts on explicit notifications: it may stay).  LinearFn takes its parameters as tensor inputs, so autograd runs their
        # post-accumulate hooks (the reducer's per-parameter notification) when LinearFn.backward returns — before a
        # queued gradient is written: the bucket would be reduced with stale contents
        # (tests/test_ddp_gloo.py::test_gradients_queued_for_a_later_launch...).  Deferral is for single-process runs.
        return False
    if not _end_flush_queued:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward_flush)
        except RuntimeError:  # not inside a backward pass
            return False
        _end_flush_queued = True
    _pending_dw.append((w, b, dy2, x2))
    if _pending_tiles() >= LINEAR_DW_TILES:
        _flush_dw(tuple(_slice_streams))
    return True


def _end_of_backward_flush() -> None:
    global _end_flush_queued
    _end_flush_queued = False
    if _pending_dw:
        _flush_dw(tuple(_slice_streams))
        SideStream.join()  # (the side streams' own end-of-backward join may already have run)


def _flush_deferred() -> None:
    if _pending_dw:
        _flush_dw(tuple(_slice_streams))


_functional.deferred_grad_flushes.append(_flush_deferred)


def _tiles_of(items: list) -> int:
    return sum(((dy2.shape[1] + 255) // 256) * ((x2.shape[1] + 255) // 256) for _, _, dy2, x2 in items)


def _pending_tiles() -> int:
    """256 x 256 output tiles of the queued weight gradients"""
    return _tiles_of(_pending_dw)


def _flush_dw(wait: tuple = ()) -> None:
    """One grouped launch for every queued weight gradient; gradients land straight in `.grad`.  `wait`: streams other
    than the current one that wrote operands (the backward's second batch slice)."""
    from .functional import grad_ready_callbacks

    if not _pending_dw:
        return
    items = list(_pending_dw)
    _pending_dw.clear()
    few = _tiles_of(items) < DW_MIN_TILES
    if few:  # too few tiles for one-tile-per-CU whole reductions: split-K GEMMs
        for w, b, dy2, x2 in items:
            SideStream.run(lambda w=w, b=b, dy2=dy2, x2=x2: _dw_db(w, b, dy2, x2), (dy2, x2), wait=wait)
        return
    # a parameter that is queued twice (shared weights) must not be written by two problems of ONE launch
    launches: list = [[]]
    seen: set = set()
    for it in items:
        if id(it[0]) in seen or len(launches[-1]) == GROUP_MAX:
            launches.append([])
            seen = set()
        seen.add(id(it[0]))
        launches[-1].append(it)

    def launch() -> None:
        for group in launches:
            probs, done = [], []
            for w, b, dy2, x2 in group:
                prms = [w] + ([b] if (b is not None and b.requires_grad) else [])
                for prm in prms:
                    if prm.grad is None:
                        prm.grad = grad_buffer(prm)
                        prm._cfhip_fresh = True
                bg, acc_b = None, False
                if len(prms) == 2 and not _SKIP_BIAS_GRAD:
                    bg, acc_b = b.grad.view(-1), not getattr(b, "_cfhip_fresh", False)
                probs.append((dy2, x2, w.grad.view(dy2.shape[1], x2.shape[1]), not getattr(w, "_cfhip_fresh", False), bg, acc_b))
                for prm in prms:
                    prm._cfhip_fresh = False
                done.extend(prms)
            ops.gemm_grouped_tn(probs)
            for prm in done:
                _functional.notify_grad_ready(prm)

    if DW_GROUP_ON_MAIN:
        for st in wait:
            _functional.cur_stream().wait_stream(st)
        launch()
    else:
        SideStream.run(launch, tuple(t for it in items for t in (it[2], it[3])), wait=wait)


# Round 4: the column reduce of the LayerNorm parameter gradients (13 us in the step, twice per block and batch slice: 0.35 ms
# of each slice's dX chain) runs on the weight-gradient lane instead: only the optimizer needs its result.  The row kernels of
# the two batch slices then write separate partial buffers and no longer wait for each other.
LN_REDUCE_ASIDE = True
SPLIT_LN_BWD = False  # round 2: ONE launch (half-wave-per-row kernel, dy and x read once) beats dx on the main stream + dgamma/dbeta on the side stream by 0.5 ms / step (profiles/r02/step_ab_ln_b128.log)


def _ln_defers(w: Tensor, b: Tensor, dy2: Tensor) -> bool:
    """LN_REDUCE_ASIDE applies: the one-launch row kernel serves this width, dgamma | dbeta are adjacent in the gradient arena
    (one reduce launch), and there is a side lane to put the reduce on"""
    return (LN_REDUCE_ASIDE and not SPLIT_LN_BWD and dy2.is_cuda and SideStream.enabled and _functional.cuda_ok()
            and w.shape[0] % 256 == 0 and w.shape[0] <= 1280 and w.grad is not None and b.grad is not None
            and b.grad.dtype == f32 and b.grad.data_ptr() == w.grad.data_ptr() + 4 * w.numel() and dy2.stride(0) % 8 == 0)


def _ln_bwd(dy2: Tensor, x2: Tensor, w: Tensor, b: Tensor, mean: Tensor, rstd: Tensor,
            dx_add: Optional[Tensor], dx_out: Optional[Tensor] = None, notify: bool = True,
            dx_add_lo: Optional[Tensor] = None, dx_lo_out: Optional[Tensor] = None) -> Tensor:
    """LayerNorm backward, split in two launches: the input gradient (with the residual-gradient add
    fused) on the main stream — it is the critical path — and dgamma / dbeta (a streaming column
    reduction straight into `.grad`) on the side stream."""
    from .functional import grad_ready_callbacks

    gamma = w.detach()
    if _ln_defers(w, b, dy2):
        dx, ws, rows = ops.layernorm_bwd_partials(dy2, x2, gamma, mean, rstd, dx_add=dx_add, dx_out=dx_out, dx_add_lo=dx_add_lo,
                                                  dx_lo_out=dx_lo_out)
        acc = not getattr(w, "_cfhip_fresh", False)
        if acc != (not getattr(b, "_cfhip_fresh", False)):
            for prm in (w, b):
                if getattr(prm, "_cfhip_fresh", False):
                    prm.grad.zero_()
            acc = True
        w._cfhip_fresh = b._cfhip_fresh = False
        d = w.shape[0]

        def reduce() -> None:
            ops.layernorm_bwd_reduce(ws, rows, d, w.grad.view(-1), b.grad.view(-1), acc)
            if notify:
                _functional.notify_grad_ready(w)
                _functional.notify_grad_ready(b)

        SideStream.run(reduce, (ws,))
        return dx

    def param_grads(with_dx: bool = False) -> Optional[Tensor]:
        for prm in (w, b):
            if prm.grad is None:
                prm.grad = grad_buffer(prm)
                prm._cfhip_fresh = True
        acc_w = not getattr(w, "_cfhip_fresh", False)
        acc_b = not getattr(b, "_cfhip_fresh", False)
        if acc_w != acc_b:
            for prm in (w, b):
                if getattr(prm, "_cfhip_fresh", False):
                    prm.grad.zero_()
            acc_w = True
        dx, _, _ = ops.layernorm_bwd(dy2, x2, gamma, mean, rstd, dgamma=w.grad.view(-1), dbeta=b.grad.view(-1),
                                     accumulate=acc_w, want_dx=with_dx, dx_add=dx_add if with_dx else None,
                                     dx_out=dx_out if with_dx else None, dx_add_lo=dx_add_lo if with_dx else None,
                                     dx_lo_out=dx_lo_out if with_dx else None)
        for prm in (w, b):
            if notify:  # (a batch-sliced backward notifies once, after the LAST slice has added its rows)
                _functional.notify_grad_ready(prm)
            else:
                prm._cfhip_fresh = False
        return dx

    if not SPLIT_LN_BWD:
        return param_grads(True)
    SideStream.run(param_grads, (dy2, x2, mean, rstd), lane=1)
    dx, _, _ = ops.layernorm_bwd(dy2, x2, gamma, mean, rstd, dx_add=dx_add, want_param_grads=False, dx_out=dx_out,
                                 dx_add_lo=dx_add_lo, dx_lo_out=dx_lo_out)
    return dx


# The forward of a block stack runs as FWD_HALVES batch-slice pipelines on separate streams (slice 0 on the caller's): every
# operator of a block is row- or sample-wise, so the slices never meet until the end of the stack, the kernels and the rows
# they see are the same (results bit-equal to one pass: tests/test_gpu_modules.py), and the LayerNorm / attention kernels of
# one slice — which leave the matrix cores idle — run beside the GEMMs of the other.  Whole step, one process, interleaved:
# batch 64 11.91 -> 11.36 ms, batch 128 20.92 -> 20.60, batch 256 40.14 -> 39.87; three slices 20.72 (tools/fwd_halves_ab.py,
# profiles/r02/fwd_halves_ab.log).  The backward is already two saturated queues (dX chain + dW GEMMs).
FWD_HALVES = int(os.environ.get("CFHIP_FWD_HALVES", "2"))


def _block_fwd(x2: Tensor, prm: tuple, num_heads: int, eps1: float, eps2: float, bsz: int, t: int,
               keep_mask: Optional[Tensor], causal: bool, quick: bool = False, streams: Optional[list] = None):
    """One pre-norm block on the [B*T, D] residual stream (f32 or bf16).  Returns (y2, saved tensors).
    `streams`: None = one pass over all rows on the current stream; a list of streams = the batch is cut into
    len(streams) contiguous slices, slice i runs on streams[i] into row slices of the same full-size tensors (every
    operator of the block is row- or sample-wise, so nothing crosses a slice)."""
    ln1_w, ln1_b, in_w, qkv_b, out_w, out_b, ln2_w, ln2_b, w1, b1, w2, b2 = prm
    d = x2.shape[1]
    fb = lambda p: None if p is None else p.detach().reshape(-1)  # noqa: E731
    in_w16, out_w16 = shadow_bf16(in_w), shadow_bf16(out_w)
    w1_16, w2_16 = shadow_bf16(w1), shadow_bf16(w2)
    m, dev = bsz * t, x2.device
    ff = w1.shape[0]
    E = lambda *shape, dt=bf16: torch.empty(shape, dtype=dt, device=dev)  # noqa: E731
    ln1, mean1, rstd1 = E(m, d), E(m, dt=f32), E(m, dt=f32)
    qkv, o2, lse = E(m, 3 * d), E(m, d), E(bsz, num_heads, t, dt=f32)
    x1 = E(m, d, dt=x2.dtype)
    ln2, mean2, rstd2 = E(m, d), E(m, dt=f32), E(m, dt=f32)
    pre, h, y = E(m, ff), E(m, ff), E(m, d, dt=x2.dtype)
    act = ops.EPI_QGELU if quick else ops.EPI_GELU
    g1, bt1, g2, bt2 = ln1_w.detach(), ln1_b.detach(), ln2_w.detach(), ln2_b.detach()

    def run(b0: int, b1_: int) -> None:
        r = slice(b0 * t, b1_ * t)
        nb = b1_ - b0
        km = None if keep_mask is None else keep_mask[b0:b1_] if keep_mask.shape[0] == bsz else keep_mask
        ops.layernorm_fwd(x2[r], g1, bt1, eps1, out=ln1[r], mean=mean1[r], rstd=rstd1[r])
        ops.gemm(ln1[r], in_w16, bias=fb(qkv_b), out=qkv[r])
        q3 = qkv[r].view(nb, t, 3 * d)
        ops.attn_fwd(q3[..., :d], q3[..., d:2 * d], q3[..., 2 * d:], num_heads, mask=km, causal=causal,
                     out=o2[r].view(nb, t, d), lse=lse[b0:b1_])
        ops.gemm(o2[r], out_w16, bias=fb(out_b), epilogue=ops.EPI_RESIDUAL, aux_in=x2[r], out=x1[r])
        ops.layernorm_fwd(x1[r], g2, bt2, eps2, out=ln2[r], mean=mean2[r], rstd=rstd2[r])
        ops.gemm(ln2[r], w1_16, bias=fb(b1), epilogue=act, aux_out=pre[r], out=h[r])
        ops.gemm(h[r], w2_16, bias=fb(b2), epilogue=ops.EPI_RESIDUAL, aux_in=x1[r], out=y[r])

    if not streams or len(streams) == 1 or bsz < 2 * len(streams):
        run(0, bsz)
    else:
        n = len(streams)
        cuts = _slice_cuts(bsz, n)
        for i, st in enumerate(streams):
            with _functional.on_stream(st):
                run(cuts[i], cuts[i + 1])
    saved = (x2, mean1, rstd1, ln1, qkv, o2, lse, x1, mean2, rstd2, ln2, pre, h, in_w16, out_w16, w1_16, w2_16)
    return y, saved


N_SAVED = 17  # tensors per block in `saved`

# Share of the batch the FIRST slice (the caller's stream, which also carries the patch embedding, the head, the loss and the
# optimizer) takes when a pass runs as two slices; 0.5 = equal halves.
FIRST_SLICE_SHARE = 0.5


def _slice_cuts(bsz: int, n: int) -> list:
    if n == 2 and FIRST_SLICE_SHARE != 0.5:
        first = min(bsz - 1, max(1, int(round(bsz * FIRST_SLICE_SHARE))))
        return [0, first, bsz]
    return [bsz * i // n for i in range(n + 1)]


# The backward of a block stack runs as BWD_HALVES batch-slice pipelines too (round 3): with the weight gradients gone
# from the per-operator path (grouped launches on their own stream) the dX chain was ONE saturated queue (19.35 of 19.95 ms
# busy on the main stream, profiles/r03) of kernels that each leave part of the chip idle (tile quantisation, LayerNorm /
# attention beside GEMMs).  Every operator of the chain is row- or sample-wise except the LayerNorm parameter gradients
# (column sums over ALL rows): slice i + 1's LayerNorm-backward launch waits for slice i's (one event) and accumulates.
BWD_HALVES = int(os.environ.get("CFHIP_BWD_HALVES", "2"))
SideStream.lanes = min(4, max(SideStream.lanes, BWD_HALVES, FWD_HALVES - 1))  # lane 0: dW / forward slice 1; lanes 1 ..: backward slices


def _block_bwd(saved: tuple, prm: tuple, num_heads: int, bsz: int, t: int, keep_mask: Optional[Tensor],
               causal: bool, d2: Tensor, quick: bool = False, streams: Optional[list] = None) -> Tensor:
    """Backward of one block: d2 = dL/dy as bf16 [B*T, D]; returns dL/dx as bf16 [B*T, D]; parameter
    gradients go straight into `.grad`.  `streams`: None = one pass over all rows on the current stream; a list = the
    batch is cut into len(streams) contiguous slices, slice i runs on streams[i] (row slices of the same full-size
    tensors).  TWO-WORD gradient stream (round 6): d2 = bf16 [2, B*T, D] — plane 0 the first word bf16(g), which every
    matrix product reads (as the reference's do: autocast rounds the f32 stream gradient where a linear layer consumes it),
    plane 1 the second word bf16(g - hi), which only the residual path carries from LayerNorm backward to LayerNorm backward
    (`cfhip_layernorm_bwd2`); the result has the same form."""
    x2, mean1, rstd1, ln1, qkv, o2, lse, x1, mean2, rstd2, ln2, pre, h, in_w16, out_w16, w1_16, w2_16 = saved
    ln1_w, ln1_b, in_w, qkv_b, out_w, out_b, ln2_w, ln2_b, w1, b1, w2, b2 = prm
    two = d2.dim() == 3
    d2_full = d2
    d2, d2_lo = (d2[0], d2[1]) if two else (d2, None)
    d = d2.shape[1]
    m = bsz * t
    E = lambda cols: torch.empty((m, cols), dtype=bf16, device=d2.device)  # noqa: E731
    # tensors that cross streams (operands of the grouped weight-gradient launch, the next block's input) are allocated
    # here, on the caller's stream, and stay alive until the end-of-backward join (_pending_dw / SideStream.keep)
    dpre, dqkv = E(pre.shape[1]), E(3 * d)
    if two:
        dx1_full, dx_full = (torch.empty((2, m, d), dtype=bf16, device=d2.device) for _ in range(2))
        (dx1, dx1_lo), (dx, dx_lo) = (dx1_full[0], dx1_full[1]), (dx_full[0], dx_full[1])
    else:
        dx1, dx = E(d), E(d)
        dx1_lo = dx_lo = None
        dx_full = dx
    qkv3, dqkv3, o3 = qkv.view(bsz, t, 3 * d), dqkv.view(bsz, t, 3 * d), o2.view(bsz, t, d)
    dact = ops.EPI_DQGELU if quick else ops.EPI_DGELU
    nsl = len(streams) if streams else 1
    if bsz < 2 * nsl:
        nsl, streams = 1, None
    cuts = _slice_cuts(bsz, nsl)
    ln_done: list = [None, None]  # event after the previous slice's LayerNorm-backward launch (LN2, LN1)

    def run(i: int) -> None:
        b0, b1_ = cuts[i], cuts[i + 1]
        r = slice(b0 * t, b1_ * t)
        nb = b1_ - b0
        last = i == nsl - 1
        km = None if keep_mask is None else keep_mask[b0:b1_] if keep_mask.shape[0] == bsz else keep_mask
        cur = _functional.cur_stream() if d2.is_cuda else None

        def ln(which: int, dy_, x_, w_, b_, mean_, rstd_, add_, out_, add_lo_=None, out_lo_=None) -> None:
            if _ln_defers(w_, b_, dy_):  # (the reduces of both slices queue up on ONE lane, in order: no event between the slices)
                _ln_bwd(dy_, x_, w_, b_, mean_, rstd_, dx_add=add_, dx_out=out_, notify=last, dx_add_lo=add_lo_, dx_lo_out=out_lo_)
                return
            if ln_done[which] is not None:
                _functional.rec_wait_event(cur, ln_done[which])
            _ln_bwd(dy_, x_, w_, b_, mean_, rstd_, dx_add=add_, dx_out=out_, notify=last, dx_add_lo=add_lo_, dx_lo_out=out_lo_)
            if nsl > 1 and not last:
                ev = torch.cuda.Event()
                _functional.rec_record_event(ev, cur)
                ln_done[which] = ev

        # channel mixing
        ops.gemm(d2[r], w2_16, b_trans=True, epilogue=dact, aux_in=pre[r], out=dpre[r])
        dln2 = ops.gemm(dpre[r], w1_16, b_trans=True)
        ln(0, dln2, x1[r], ln2_w, ln2_b, mean2[r], rstd2[r], d2[r], dx1[r], None if d2_lo is None else d2_lo[r],
           None if dx1_lo is None else dx1_lo[r])
        # token mixing
        d_o = ops.gemm(dx1[r], out_w16, b_trans=True)
        q3, dq3 = qkv3[b0:b1_], dqkv3[b0:b1_]
        # (the two attention passes are independent kernels — `parts` — but running them on two streams measured
        # slower end-to-end: the chain has to wait for both anyway)
        ops.attn_bwd(q3[..., :d], q3[..., d:2 * d], q3[..., 2 * d:], o3[b0:b1_], d_o.view(nb, t, d), lse[b0:b1_], num_heads,
                     dq=dq3[..., :d], dk=dq3[..., d:2 * d], dv=dq3[..., 2 * d:], mask=km, causal=causal)
        dln1 = ops.gemm(dqkv[r], in_w16, b_trans=True)
        ln(1, dln1, x2[r], ln1_w, ln1_b, mean1[r], rstd1[r], dx1[r], dx[r], None if dx1_lo is None else dx1_lo[r],
           None if dx_lo is None else dx_lo[r])

    if nsl == 1:
        run(0)
    else:
        for i, st in enumerate(streams):
            with _functional.on_stream(st):
                run(i)
    # parameter gradients: queued for the grouped launch (or issued on the side stream — round-1/2 path); whoever
    # launches them waits for every slice stream
    _queue_dw(w2, b2, d2, h)
    _queue_dw(w1, b1, dpre, ln2)
    _queue_dw(out_w, out_b, dx1, o2)
    _queue_dw(in_w, qkv_b, dqkv, ln1)
    if two:
        SideStream.keep.append(d2_full)  # (its planes are views: the whole tensor stays until the end-of-backward join)
    return dx_full


# The residual-gradient stream of a block stack whose residual stream is f32: 2 (default since round 6) = two bf16 words ~ f32, the
# reference's autocast semantics (the gradient of an f32 tensor is f32, only the matrix products see it rounded;
# `cfhip_layernorm_bwd2`); 1 = one bf16 word (rounds 1-5).  A bf16 residual stream (the UNet's transformer tokens: bf16 in the
# reference too) always has a one-word gradient.  A stack can ask for either through its metas
# (modules.MixedStackedEncoder.grad_stream_words).  What it buys, measured against the reference's OWN bf16-autocast distance from
# fp32 (tests/golden/*_yardstick.pt): ViT-B/16 x 8, 152 gradients: one word max 1.31 x / median 1.10 x, two words max 1.09 x /
# median 0.96 x; CLIP b32 head token 1.51 x -> 0.6 x.  What it costs: 4 B / element more in each LayerNorm backward,
# ViT-B/16 b128 17.45 -> 17.79 ms (tools/grad_words_ab.py, profiles/r06/grad_words_ab.txt).  DESIGN §3.6b.
GRAD_STREAM_WORDS = int(os.environ.get("CFHIP_GRAD_STREAM_WORDS", "2"))


def _as_bf16_rows(dy: Tensor, rows: int, d: int, words: int = 1) -> Tensor:
    """the incoming gradient as bf16 rows [rows, d]; `words` = 2: [2, rows, d] — first and second word (an f32 gradient is split
    exactly into hi + lo; a bf16 one has no second word: zeros)"""
    if words == 2:
        out = torch.empty((2, rows, d), dtype=bf16, device=dy.device)
        if dy.dtype == bf16:
            out[0].copy_(dy.contiguous().view(rows, d))
            out[1].zero_()
        else:
            ops.split_f32(dy.float().contiguous().view(rows, d), out[0], out[1])
        return out
    if dy.dtype != bf16:
        dy = ops.to_bf16(dy.float().contiguous())
    return dy.contiguous().view(rows, d)


class MixingBlockFn(Function):
    @staticmethod
    def forward(ctx: Any, x: Tensor, ln1_w: Tensor, ln1_b: Tensor, in_w: Tensor, qkv_b: Optional[Tensor],
                out_w: Tensor, out_b: Optional[Tensor], ln2_w: Tensor, ln2_b: Tensor, w1: Tensor,
                b1: Optional[Tensor], w2: Tensor, b2: Optional[Tensor], num_heads: int, eps1: float,
                eps2: float, keep_mask: Optional[Tensor], causal: bool, quick: bool = False) -> Tensor:
        bsz, t, d = x.shape
        if x.dtype not in (bf16, f32):
            x = x.float()
        x2 = x.contiguous().view(bsz * t, d)  # residual stream: f32 (reference autocast semantics) or bf16
        prm = (ln1_w, ln1_b, in_w, qkv_b, out_w, out_b, ln2_w, ln2_b, w1, b1, w2, b2)
        y, saved = _block_fwd(x2, prm, num_heads, eps1, eps2, bsz, t, keep_mask, causal, quick)
        ctx.save_for_backward(*saved, keep_mask)
        ctx.params = prm
        ctx.meta = (bsz, t, d, num_heads, causal, quick)
        return y.view(bsz, t, d)

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        *saved, keep_mask = ctx.saved_tensors
        bsz, t, d, num_heads, causal, quick = ctx.meta
        d2 = _as_bf16_rows(dy, bsz * t, d)
        dx = _block_bwd(tuple(saved), ctx.params, num_heads, bsz, t, keep_mask, causal, d2, quick)
        _flush_dw()
        return (dx.view(bsz, t, d),) + (None,) * 18


# ---- launch plans (round 4) -------------------------------------------------------------------------------
# The host pays ~20 us of Python per kernel launch of a block stack (argument checks, torch.empty, stream contexts, ctypes
# marshalling: 10 ms of a ViT-B/16 step's 18, 17.5 of the CLIP step's 18.8 — which is therefore bound by the host).  A stack
# called again with the same shapes issues the SAME launches on the SAME streams; only addresses could differ.  So: the
# second call of a stack is RECORDED at the C-ABI boundary (`_lib.RECORDER`: function + raw argument tuple of every
# cfhip_* call, plus the stream waits / event records between the batch slices and the gradient notifications), every
# tensor allocated while recording is kept, and later calls REPLAY the list: the same cfhip_* calls with the same
# arguments, ~2 us each, no wrappers, no allocations.  Still eager launches on the same streams — not a hipGraph (whose replay
# costs ~25 us per node on this ROCm) — and bit-identical by construction (same kernels, same arguments).
#   * inputs (x in forward, dy in backward) that arrive at another address are copied into the recorded buffer (one D2D copy);
#   * the bf16 weight shadows alternate between two arenas when the optimizer updates inside backward: the plan carries the
#     address map and a twin argument list;
#   * anything the recording did not see — another shape, a mask at another address, gradient accumulation state, timers or
#     FLOP counters attached, a hipGraph capture, a second forward before the backward of the first — takes the normal path;
#   * ALIASING CONTRACT (as with the static outputs of a captured graph): the output of a replayed forward and the input
#     gradient of a replayed backward ARE the recorded buffers — valid until the next training-mode forward / backward of the
#     same stack overwrites them.  Everything inside a step (the head, the loss, autograd) consumes them before that; code
#     that keeps a stack's training-mode output across steps must clone it (or run with CFHIP_STACK_PLANS=0).
STACK_PLANS = os.environ.get("CFHIP_STACK_PLANS", "1") != "0"
_plans: dict = {}  # id(first parameter) -> StackPlan
_plan_conflicts: dict = {}  # id(first parameter) -> times a forward found the stack's plan still in flight
_PLAN_CACHE = 8


class StackPlan:
    def __init__(self, key: tuple, params: tuple) -> None:
        self.key = key
        self.calls = 0            # forward calls seen with this key (the first one runs unrecorded: lazy initialisations)
        self.fwd: list = []       # recorded (kind, fn, args)
        self.bwd: list = []
        self.fwd_alt: Optional[list] = None  # the same with the weight shadows of the other arena
        self.bwd_alt: Optional[list] = None
        self.keep: list = []      # every tensor allocated while recording
        self.x_in: Optional[Tensor] = None
        self.y_out: Optional[Tensor] = None
        self.dy_in: Optional[Tensor] = None
        self.dx_out: Optional[Tensor] = None
        self.all_saved: Optional[list] = None
        self.w_ptr = 0            # address of the first weight shadow at recording time
        self.w_alt = 0
        self.in_flight = False    # forward replayed / recorded, backward not yet run
        self.ready_fwd = False
        self.ready_bwd = False
        self.x_live: Optional[Tensor] = None  # the caller's input of the forward in flight (replayed launches read it in place)
        self.x_sites: list = []
        self.dy_sites: list = []
        self.x_now = self.dy_now = 0
        self.disabled = False     # a usage the plan cannot follow was seen (two forwards before a backward): normal path from then on
        self.mask_tensor: Optional[Tensor] = None
        self.state = key[-1]      # (1, fingerprint): every gradient slot of the stack is written first (lazy zero); (0, ...): every one is accumulated into; fingerprint = where parameters and gradients live
        self.params_ref = [__import__("weakref").ref(p) for p in params if p is not None][:1]

    @staticmethod
    def _subst(ops_: list, mp: dict) -> list:
        out = []
        for kind, f, a in ops_:
            if kind == 0:
                a = tuple(mp.get(v, v) if type(v) is int else v for v in a)
            out.append((kind, f, a))
        return out

    def _patch_sites(self, base: int, nbytes: int) -> list:
        """(list id, entry index, argument index, byte offset) of every recorded pointer argument inside [base, base + nbytes)"""
        sites = []
        for li, ops_ in enumerate((self.fwd, self.bwd)):
            for ei, (kind, _f, a) in enumerate(ops_):
                if kind == 3:
                    # the host problem table of a grouped weight-gradient launch (ops.gemm_grouped_tn): its A / B operand
                    # addresses are not in any argument tuple.  Site = (table row, field name); the table object is shared
                    # by the twin lists (no weight shadow is an operand of a weight gradient), so it is patched in place.
                    for pi in range(a):
                        for field in ("A", "B"):
                            v = getattr(_f[pi], field) or 0
                            if base <= v < base + nbytes:
                                sites.append((li, ei, (pi, field), v - base))
                    continue
                if kind != 0:
                    continue
                for ai, v in enumerate(a):
                    if type(v) is int and base <= v < base + nbytes:
                        sites.append((li, ei, ai, v - base))
        return sites

    def repoint(self, sites: list, new_base: int) -> None:
        """the stack's input (or incoming gradient) lives at another address this step: rewrite the few launches that read it —
        in both twin lists — instead of copying the tensor into the recorded buffer"""
        for li, ei, ai, off in sites:
            if type(ai) is tuple:  # a row of a grouped launch's problem table (shared by both twin lists)
                setattr((self.fwd if li == 0 else self.bwd)[ei][1][ai[0]], ai[1], new_base + off)
                continue
            for ops_ in ((self.fwd, self.fwd_alt) if li == 0 else (self.bwd, self.bwd_alt)):
                if ops_ is None:
                    continue
                kind, f, a = ops_[ei]
                ops_[ei] = (kind, f, a[:ai] + (new_base + off,) + a[ai + 1:])

    def finish_recording(self, params: tuple) -> None:
        """after the recorded backward: the twin lists for the other shadow arena, the launches that read the inputs"""
        self.x_sites = self._patch_sites(self.x_in.data_ptr(), self.x_in.numel() * self.x_in.element_size())
        self.dy_sites = self._patch_sites(self.dy_in.data_ptr(), self.dy_in.numel() * self.dy_in.element_size())
        self.x_now, self.dy_now = self.x_in.data_ptr(), self.dy_in.data_ptr()
        mp = {}
        for p in params:
            ar = getattr(p, "_cfhip_arena", None) if p is not None else None
            views = getattr(ar, "_shadow_views", None) if ar is not None else None
            if views is None:
                continue
            i = ar._index(p)
            a0, a1 = views[0][i].data_ptr(), views[1][i].data_ptr()
            mp[a0], mp[a1] = a1, a0
        if mp:
            self.fwd_alt, self.bwd_alt = self._subst(self.fwd, mp), self._subst(self.bwd, mp)
            self.w_alt = mp.get(self.w_ptr, 0)


def _replay(ops_: list) -> None:
    from . import _lib

    for kind, f, a in ops_:
        if kind == 0:
            rc = f(*a)
            if rc:
                _lib.check(rc, "launch plan")
        elif kind == 1:
            f(*a)
        elif kind == 3:
            continue  # the problem table of the grouped launch that follows (kept for repoint)
        else:  # gradient notification, on the stream it fired on
            with _functional.on_stream(a):
                _functional.notify_grad_ready(f)


class _KeepAllocations:
    """while recording: every tensor torch.empty / empty_like hands out stays alive with the plan (its address is in the
    recorded arguments).  The wrappers allocate with these two calls only (no fills: nothing an ATen kernel would have to redo).
    Two module attributes are swapped for the duration of ONE recorded forward / backward; only allocations made by the
    RECORDING thread are kept (a loader thread's pinned buffers allocated in that window pass through untouched)."""

    def __init__(self, keep: list) -> None:
        self.keep = keep

    def __enter__(self) -> None:
        import threading

        self.empty, self.empty_like = torch.empty, torch.empty_like
        keep, e0, e1 = self.keep, self.empty, self.empty_like
        me, ident = threading.get_ident(), threading.get_ident

        def empty(*a, **k):
            t = e0(*a, **k)
            if ident() == me:
                keep.append(t)
            return t

        def empty_like(*a, **k):
            t = e1(*a, **k)
            if ident() == me:
                keep.append(t)
            return t

        torch.empty, torch.empty_like = empty, empty_like

    def __exit__(self, *exc) -> None:
        torch.empty, torch.empty_like = self.empty, self.empty_like


def _grad_state(params: tuple) -> Optional[tuple]:
    """(state, fingerprint) or None.  state 1: every parameter gradient of the stack is 'fresh' (the first kernel writes it:
    lazy zero-grad); 0: none is (zeroed arena, kernels accumulate); None: mixed, frozen weights, or a gradient that does not
    exist yet.  fingerprint: a checksum of where every parameter and every gradient LIVES — the recorded launches carry those
    addresses, so a `.grad` somebody replaced, parameters re-homed into another arena or `p.data = ...` must give another plan
    (a new recording), never a replay into the old buffers."""
    n_fresh = n = 0
    fp = 0
    for p in params:
        if p is None:
            continue
        g = p.grad
        if g is None or not p.requires_grad:
            return None
        n += 1
        n_fresh += 1 if getattr(p, "_cfhip_fresh", False) else 0
        fp = (fp * 1000003 + p.data_ptr() + 31 * g.data_ptr()) & 0xFFFFFFFFFFFFFFFF
    state = 1 if n_fresh == n else 0 if n_fresh == 0 else None
    return None if state is None else (state, fp)


def _plan_for(x: Tensor, metas: tuple, keep_mask: Optional[Tensor], causal: bool, params: tuple, training: bool) -> Optional[StackPlan]:
    """the plan of this stack call, or None when plans do not apply to it (`training`: a backward pass will follow)"""
    from . import _lib

    if (not STACK_PLANS or not x.is_cuda or not training or ops.GEMM_TIMER is not None or ops.FLOP_COUNTER is not None
            or _lib.RECORDER is not None or torch.cuda.is_current_stream_capturing() or params[2] is None):
        return None
    state = _grad_state(params)
    if state is None:
        return None  # gradients outside an arena / frozen weights / some slots written and others accumulated: the normal path
    key = (tuple(x.shape), x.dtype, metas, causal, None if keep_mask is None else keep_mask.data_ptr(), FWD_HALVES, BWD_HALVES,
           DW_GROUP_BLOCKS, DW_GROUP_TILES, DW_TAIL_BLOCKS, DW_PEROP_TAIL, LN_REDUCE_ASIDE, DW_GROUP_ON_MAIN, FIRST_SLICE_SHARE, FUSE_BIAS_GRAD,
           ops.FORCE_SPLIT_K, tuple(id(cb) for cb in _functional.grad_ready_callbacks), x.requires_grad, GRAD_STREAM_WORDS, state)
    pid = id(params[0])
    plan = _plans.get(pid)
    if plan is None or plan.key != key or (plan.params_ref and plan.params_ref[0]() is not params[0]):
        if len(_plans) >= _PLAN_CACHE:
            _plan_conflicts.pop(next(iter(_plans)), None)  # (the conflict count lives and dies with its plan: ids are reused)
            _plans.pop(next(iter(_plans)))
        if plan is not None and plan.params_ref and plan.params_ref[0]() is not params[0]:
            _plan_conflicts.pop(pid, None)  # another model now owns this id
        plan = _plans[pid] = StackPlan(key, params)
    return plan


class MixingStackFn(Function):
    """ALL blocks of a `MixedStackedEncoder` as one autograd node: the residual-gradient stream stays
    bf16 from block to block (one node per block makes autograd cast every block's bf16 input gradient
    to the f32 of the forward activation, and the next backward cast it back: 2 x 12 elementwise
    kernels per step), and the engine walks 1 node instead of 12."""

    @staticmethod
    def _slices(metas: tuple, default: int) -> int:
        """batch slices of this stack: the 5th meta entry when the owner set one (modules.MixedStackedEncoder.stack_slices — CLIP runs
        its two towers side by side on two streams instead of two slices of one tower after the other), else the module default"""
        m0 = metas[0]
        return int(m0[4]) if len(m0) > 4 and m0[4] else default

    @staticmethod
    def _slices_bwd(metas: tuple, default: int) -> int:
        m0 = metas[0]
        return int(m0[5]) if len(m0) > 5 and m0[5] else MixingStackFn._slices(metas, default)

    @staticmethod
    def _words(metas: tuple, x_is_f32: bool) -> int:
        """words of the residual-gradient stream (7th meta entry, else the module default); two only under an f32 stream"""
        m0 = metas[0]
        w = int(m0[6]) if len(m0) > 6 and m0[6] else GRAD_STREAM_WORDS
        return 2 if (w == 2 and x_is_f32) else 1

    @staticmethod
    def _forward_body(cur: Tensor, bsz: int, t: int, metas: tuple, keep_mask: Optional[Tensor], causal: bool, params: tuple):
        nblk = len(metas)
        all_saved = []
        streams = None
        FWD_HALVES = MixingStackFn._slices(metas, globals()["FWD_HALVES"])
        if FWD_HALVES > 1 and cur.is_cuda and bsz >= 2 * FWD_HALVES:
            # every bf16 weight shadow a slice will read is (re)cast HERE, on the caller's stream, BEFORE the side streams
            # fork: a stale shadow (weights stepped by a torch optimizer, first forward without a ParamArena) would
            # otherwise be re-cast on the main stream after the fork while the side-stream slice already reads it
            # (ADVICE r2, high).  With fresh shadows the calls below are attribute reads.
            for i in range(nblk):
                for j in (2, 4, 8, 10):  # in_w, out_w, w1, w2
                    shadow_bf16(params[12 * i + j])
            # slice 0 stays on the caller's stream, the others take side streams that have waited for it; every slice
            # is an independent pipeline through all blocks, joined once at the end of the stack
            main = _functional.cur_stream()
            streams = [main] + [SideStream.fork(lane) for lane in range(FWD_HALVES - 1)]
            if any(st is None for st in streams):
                streams = None
        for i, meta in enumerate(metas):
            num_heads, eps1, eps2 = meta[:3]
            quick = bool(meta[3]) if len(meta) > 3 else False
            cur, saved = _block_fwd(cur, params[12 * i:12 * i + 12], num_heads, eps1, eps2, bsz, t, keep_mask, causal,
                                    quick, streams)
            all_saved.extend(saved)
        if streams is not None:
            for st in streams[1:]:
                _functional.rec_wait_stream(_functional.cur_stream(), st)
        return cur, all_saved

    @staticmethod
    def forward(ctx: Any, x: Tensor, metas: tuple, keep_mask: Optional[Tensor], causal: bool,
                *params: Optional[Tensor]) -> Tensor:
        from . import _lib

        bsz, t, d = x.shape
        if x.dtype not in (bf16, f32):
            x = x.float()
        cur = x.contiguous().view(bsz * t, d)
        assert len(params) == 12 * len(metas)
        ctx.params = params
        ctx.meta = (bsz, t, d, metas, causal)
        ctx.words = MixingStackFn._words(metas, x.dtype == f32)
        ctx.plan = None
        plan = _plan_for(x, metas, keep_mask, causal, params, any(ctx.needs_input_grad))
        if plan is not None and plan.in_flight and not plan.disabled:
            # a second forward of this stack before the backward of the first: the recorded buffers would be overwritten under
            # the pending backward.  The plan in flight is dropped from the table (the pending backward — if one ever comes —
            # still holds it through its ctx and replays on its own buffers; a forward that never gets a backward, e.g. an
            # exception mid-step or an evaluation with grad enabled, releases it with its graph) and the stack starts a NEW
            # plan: this call runs the normal path, the next one records.  A stack that keeps doing it (towers shared inside
            # one step) would re-record every other step: after the third time it keeps the normal path for good, said once.
            pid = id(params[0])
            n = _plan_conflicts[pid] = _plan_conflicts.get(pid, 0) + 1
            plan = _plans[pid] = StackPlan(plan.key, params)
            if n > 2:
                plan.disabled = True
                if n == 3:
                    import warnings

                    warnings.warn("cfhip: a block stack is called again before the backward of its previous call, repeatedly: "
                                  "launch plans are off for this stack (normal launch path; results are the same)")
        if plan is not None and not plan.in_flight and not plan.disabled:
            plan.calls += 1
            plan.mask_tensor = keep_mask
            # every weight shadow the recorded launches read is brought up to date HERE (attribute reads when the optimizer
            # keeps them fresh; a cast on the caller's stream, before the replay forks, when a torch optimizer stepped the
            # parameter or nobody re-cast it between two accumulation micro-batches: the recording may not hold that cast —
            # ADVICE r4, medium)
            for j in range(2, len(params), 12):
                for q in (params[j], params[j + 2], params[j + 6], params[j + 8]):  # in_w, out_w, w1, w2
                    shadow_bf16(q)
            w_now = shadow_bf16(params[2]).data_ptr()
            if plan.ready_fwd and plan.ready_bwd and w_now in (plan.w_ptr, plan.w_alt):
                # ---- replay
                if cur.data_ptr() != plan.x_now:
                    plan.repoint(plan.x_sites, cur.data_ptr())
                    plan.x_now = cur.data_ptr()
                plan.x_live = cur  # (block 0's backward reads it again: alive until then)
                _replay(plan.fwd if w_now == plan.w_ptr else plan.fwd_alt)
                plan.in_flight = True
                ctx.plan = plan
                ctx.plan_alt = w_now != plan.w_ptr
                return plan.y_out.detach().view(bsz, t, d)
            if plan.calls >= 2 and not plan.ready_fwd:
                # ---- record (the first call ran the normal path: side streams, shadows and lazy state exist by now)
                plan.keep, plan.fwd, plan.bwd = [], [], []
                plan.w_ptr = w_now
                _lib.RECORDER = plan.fwd
                try:
                    with _KeepAllocations(plan.keep):
                        plan.x_in = cur
                        y, all_saved = MixingStackFn._forward_body(cur, bsz, t, metas, keep_mask, causal, params)
                finally:
                    _lib.RECORDER = None
                plan.y_out, plan.all_saved = y, all_saved
                plan.ready_fwd, plan.in_flight = True, True
                ctx.plan, ctx.plan_alt = plan, False
                return y.detach().view(bsz, t, d)
        y, all_saved = MixingStackFn._forward_body(cur, bsz, t, metas, keep_mask, causal, params)
        ctx.save_for_backward(*all_saved, keep_mask)
        return y.view(bsz, t, d)

    @staticmethod
    def _backward_body(all_saved: list, keep_mask: Optional[Tensor], bsz: int, t: int, metas: tuple, causal: bool, d2: Tensor,
                       params: tuple) -> Tensor:
        streams = None
        BWD_HALVES = MixingStackFn._slices_bwd(metas, globals()["BWD_HALVES"])
        if BWD_HALVES > 1 and d2.is_cuda and bsz >= 2 * BWD_HALVES:
            main = _functional.cur_stream()
            # lane 1, 2, ...: lane 0 is the stream of the weight-gradient launches
            streams = [main] + [SideStream.fork(lane + 1) for lane in range(BWD_HALVES - 1)]
            if any(st is None for st in streams) or len({id(st) for st in streams}) != len(streams) or SideStream.get(0) in streams[1:]:
                streams = None
        _slice_streams[:] = streams[1:] if streams else []
        try:
            for i in range(len(metas) - 1, -1, -1):
                saved = tuple(all_saved[N_SAVED * i:N_SAVED * (i + 1)])
                quick = bool(metas[i][3]) if len(metas[i]) > 3 else False
                before = _pending_tiles()
                global _perop_now
                if i < DW_PEROP_TAIL and not _perop_now:
                    _flush_dw(tuple(_slice_streams))  # what the earlier blocks queued goes out first
                    _perop_now = True
                d2 = _block_bwd(saved, params[12 * i:12 * i + 12], metas[i][0], bsz, t, keep_mask, causal, d2, quick, streams)
                if DW_GROUP_BLOCKS > 0 and DW_GROUP_TILES > 0:
                    now = _pending_tiles()
                    # one more block like this one would start a second round — or: the pass is about to end (DW_TAIL_BLOCKS)
                    if now + (now - before) > DW_GROUP_TILES or i <= DW_TAIL_BLOCKS:
                        _flush_dw(tuple(_slice_streams))
                elif DW_GROUP_BLOCKS > 0 and (len(metas) - i) % DW_GROUP_BLOCKS == 0:
                    _flush_dw(tuple(_slice_streams))
            _flush_dw(tuple(_slice_streams))
            if streams is not None:
                for st in streams[1:]:
                    _functional.rec_wait_stream(_functional.cur_stream(), st)
                SideStream.keep.append(d2)  # written by both slice streams, allocated on the caller's
            if d2.dim() == 3:  # two-word stream: what leaves the stack is the f32 gradient of its f32 input, hi + lo
                d2 = ops.join_bf16x2(d2[0], d2[1])
        except BaseException:
            _pending_dw.clear()  #  nothing queued by a failed pass may be flushed into `.grad` by the next one
            raise
        finally:
            _slice_streams[:] = []
            _perop_now = False
        return d2

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        from . import _lib

        bsz, t, d, metas, causal = ctx.meta
        params = ctx.params
        plan: Optional[StackPlan] = ctx.plan
        nret = (None, None, None) + (None,) * len(params)
        if plan is None:
            *all_saved, keep_mask = ctx.saved_tensors
            d2 = _as_bf16_rows(dy, bsz * t, d, ctx.words)
            d2 = MixingStackFn._backward_body(list(all_saved), keep_mask, bsz, t, metas, causal, d2, params)
            return (d2.view(bsz, t, d),) + nret
        plan.in_flight = False
        if _plan_conflicts:
            # a clean forward / backward pair: an occasional forward-without-backward (a periodic evaluation with grad enabled)
            # must not add up to "this stack keeps doing it" over a long run
            _plan_conflicts.pop(id(params[0]), None)
        km = plan.mask_tensor
        fresh = _grad_state(params) == plan.state  # the write / accumulate flags in the recorded arguments still apply
        if plan.ready_bwd and fresh and _lib.RECORDER is None and ops.GEMM_TIMER is None and ops.FLOP_COUNTER is None:
            # ---- replay
            d2 = _as_bf16_rows(dy, bsz * t, d, ctx.words)
            if d2.data_ptr() != plan.dy_now:
                plan.repoint(plan.dy_sites, d2.data_ptr())
                plan.dy_now = d2.data_ptr()
            SideStream.queue_join()
            _replay(plan.bwd_alt if ctx.plan_alt else plan.bwd)
            SideStream.keep.append(d2)  # read by launches on the side lanes: alive until the end-of-backward join
            plan.x_live = None
            return (plan.dx_out.detach().view(bsz, t, d),) + nret
        if not plan.ready_bwd and fresh and _lib.RECORDER is None:
            # ---- record the backward of the recorded forward (the incoming gradient is converted BEFORE the recording
            # starts: that launch reads this pass's autograd buffer, which no later pass will see at that address)
            d2 = _as_bf16_rows(dy, bsz * t, d, ctx.words)
            plan.dy_in = d2
            _lib.RECORDER = plan.bwd
            try:
                with _KeepAllocations(plan.keep):
                    dx = MixingStackFn._backward_body(plan.all_saved, km, bsz, t, metas, causal, d2, params)
            finally:
                _lib.RECORDER = None
            plan.dx_out = dx
            plan.keep.append(dx)
            plan.finish_recording(params)
            plan.ready_bwd = True
            return (dx.detach().view(bsz, t, d),) + nret
        # the plan cannot serve this backward (gradient accumulation state, timers attached): the normal path on its buffers
        d2 = _as_bf16_rows(dy, bsz * t, d, ctx.words)
        saved = list(plan.all_saved)
        if plan.x_live is not None:
            saved[0] = plan.x_live  # (a replayed forward read the caller's input in place, not the recorded buffer)
            plan.x_live = None
        d2 = MixingStackFn._backward_body(saved, km, bsz, t, metas, causal, d2, params)
        return (d2.view(bsz, t, d),) + nret


def mixing_block(x: Tensor, *args: Any) -> Tensor:
    return MixingBlockFn.apply(x, *args)


def mixing_stack(x: Tensor, metas: tuple, keep_mask: Optional[Tensor], causal: bool, params: list) -> Tensor:
    """`metas[i]` = (num_heads, eps1, eps2[, quick_gelu]) of block i, `params` = its 12 parameters, concatenated."""
    return MixingStackFn.apply(x, metas, keep_mask, causal, *params)
