"""Tensor-level wrappers over the C-ABI (include/cfhip.h).  No autograd here.

Every function takes / returns torch tensors that live on the HIP device, launches on torch's
current stream and never synchronises.  PyTorch is used only as the owner of device memory.
"""
import math
import os
from typing import Optional, Tuple

import ctypes

import torch
from torch import Tensor

from . import _lib

EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_QGELU, EPI_DQGELU = 0, 1, 2, 3, 4, 5

bf16 = torch.bfloat16
f32 = torch.float32


# The raw handle of torch's current HIP stream.  `torch.cuda.current_stream().cuda_stream` builds a Stream object through
# four Python layers (9 us per call: 768 + ~800 calls = 14 ms of the host's 65 ms per UNet 64^2 step, tools/host_profile.py);
# the two C entry points behind it take ~0.3 us.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: Tensor, dtype: torch.dtype, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"cfhip: `{name}` must live on the HIP device (got {t.device})")
    if t.dtype != dtype:
        raise TypeError(f"cfhip: `{name}` must be {dtype}, got {t.dtype}")


def _mat(t: Tensor, name: str) -> Tuple[int, int, int]:
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"cfhip: `{name}` must be 2-D with a contiguous last dim, got {tuple(t.shape)} / {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


def set_option(name: str, value: int) -> None:
    """process-wide tuning knob of libcfhip.so (see cfhip_set_option in include/cfhip.h)"""
    _lib.check(_lib.load().cfhip_set_option(name.encode(), int(value)), "set_option")


# ---------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------


FORCE_SPLIT_K = 0  # A/B runs: > 0 = every weight-gradient GEMM takes this split


def pick_split_k(m: int, n: int, k: int) -> int:
    """Split the reduction when the output has too few 128x128 tiles to fill 256 CUs."""
    if FORCE_SPLIT_K > 0:
        return FORCE_SPLIT_K
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    steps = (k + 63) // 64
    if tiles >= 256 or steps < 16:
        return 1
    # (deeper than 32: the reduce pass costs more than it fills.  Round 6 tried 64-128 slices aimed at 1 024 workgroups for long reductions
    # over <= 16 tiles: 256^2 x 1 UNet step -5 ms, level at 64^2 x 8 — and nothing once such reductions of >= 49 152 rows go to the grouped
    # whole-K launches (fused.LINEAR_DW_MIN_ROWS), which is worth as much: profiles/r06/unet_split_deep_ab.txt; removed)
    split = min(steps // 4, (512 + tiles - 1) // tiles, 32)
    return max(1, split)


class GemmTimer:
    """Opt-in (bench.py's roofline pass): HIP-event pairs around every GEMM launch of the REAL training step, recorded
    on the stream the kernel is launched on (torch's current stream at the call: the main stream or a side stream),
    keyed by (layout, M, N, K, epilogue).  Off by default: `ops.GEMM_TIMER is None` costs one attribute read."""

    def __init__(self) -> None:
        self.records: list = []

    def durations(self) -> dict:
        """key -> [count, total seconds]; synchronises on the recorded events.  A launch that took more than 8 x the median of
        its key (one 100 ms hiccup among 72 launches of a 150 us kernel was seen once: a profile step that had to grow the
        allocator) counts with the median instead — the table describes the steady state."""
        per: dict = {}
        for key, e0, e1 in self.records:
            e1.synchronize()
            per.setdefault(key, []).append(e0.elapsed_time(e1) * 1.0e-3)
        out: dict = {}
        for key, ds in per.items():
            med = sorted(ds)[len(ds) // 2]
            out[key] = [len(ds), sum(d if d <= 8.0 * med else med for d in ds)]
        return out


GEMM_TIMER: Optional[GemmTimer] = None


class FlopCounter:
    """Opt-in (bench.py --workload unet | clip): algorithmic MFMA-class FLOPs of the launches of a step, by family.
    GEMM 2*M*N*K; implicit 3x3 convolution 2 * pixels * Cout * 9 * Cin (forward, input gradient = the same kernel,
    weight gradient); attention 4 * B * H * Tq * Tk * dh forward and 2.5x that backward (5 products: the algorithmic
    count, not the 7 the two-pass backward executes)."""

    def __init__(self) -> None:
        self.flops = {"gemm": 0.0, "conv3x3": 0.0, "attention": 0.0}

    def total(self) -> float:
        return sum(self.flops.values())


FLOP_COUNTER: Optional[FlopCounter] = None


def gemm(
    a: Tensor,
    b: Tensor,
    *,
    a_trans: bool = False,
    b_trans: bool = False,
    bias: Optional[Tensor] = None,
    epilogue: int = EPI_NONE,
    aux_in: Optional[Tensor] = None,
    aux_out: Optional[Tensor] = None,
    out: Optional[Tensor] = None,
    out_dtype: torch.dtype = bf16,
    accumulate: bool = False,
    split_k: int = 1,
    bias_grad: Optional[Tensor] = None,
    bias_grad_accumulate: bool = False,
) -> Tensor:
    """C[m,n] = epilogue(sum_k A(m,k) B(n,k)); see cfhip_gemm_bf16 in include/cfhip.h.
    `bias_grad` (f32 [M], layout a_trans & b_trans only): (+)= sum_k A(m,k), i.e. colsum(dY) of a dW GEMM."""
    # (this wrapper runs 400-500 times per UNet / CLIP step on the thread that issues every launch: shapes and strides are
    # read once as tuples, the checks are inline — tools/host_profile.py had it at 10 us per call, a tenth of the host's step)
    if a.dtype is not bf16 or b.dtype is not bf16 or not a.is_cuda or not b.is_cuda:
        _need(a, bf16, "a")
        _need(b, bf16, "b")
    sha, shb, sta, stb = a.shape, b.shape, a.stride(), b.stride()
    if len(sha) != 2 or sta[1] != 1:
        _mat(a, "a")
    if len(shb) != 2 or stb[1] != 1:
        _mat(b, "b")
    lda, ldb = sta[0], stb[0]
    m, k = (sha[1], sha[0]) if a_trans else (sha[0], sha[1])
    n, kb = (shb[1], shb[0]) if b_trans else (shb[0], shb[1])
    if k != kb:
        raise ValueError(f"cfhip gemm: reduction dims differ ({k} vs {kb})")
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=a.device)
        ldc, odt = n, out_dtype
    else:
        odt, sho, sto = out.dtype, out.shape, out.stride()
        if not out.is_cuda or (odt is not bf16 and odt is not f32) or len(sho) != 2 or sho[0] != m or sho[1] != n or sto[1] != 1:
            raise ValueError("cfhip gemm: bad `out`")
        ldc = sto[0]
    if bias is not None and (bias.dtype is not f32 or not bias.is_cuda or bias.numel() != n or not bias.is_contiguous()):
        _need(bias, f32, "bias")
        raise ValueError("cfhip gemm: bias must be a contiguous f32 [N]")
    if aux_in is not None or aux_out is not None:
        for t, nm in ((aux_in, "aux_in"), (aux_out, "aux_out")):
            if t is not None:
                # the residual operand follows the output dtype (f32 residual stream); everything else is bf16
                _need(t, f32 if (nm == "aux_in" and epilogue == EPI_RESIDUAL and odt is f32) else bf16, nm)
                stt = t.stride()
                if tuple(t.shape) != (m, n) or stt[1] != 1 or stt[0] != ldc:
                    raise ValueError(f"cfhip gemm: `{nm}` must match the output layout")
    if bias_grad is not None:
        _need(bias_grad, f32, "bias_grad")
        if bias_grad.numel() != m or not bias_grad.is_contiguous():
            raise ValueError("cfhip gemm: bias_grad must be a contiguous f32 [M]")
    ws, ws_bytes = None, 0
    if split_k > 1:
        ws = torch.empty((split_k * m * (n + (1 if bias_grad is not None else 0)),), dtype=f32, device=a.device)
        ws_bytes = ws.numel() * 4
    if FLOP_COUNTER is not None:
        FLOP_COUNTER.flops["gemm"] += 2.0 * m * n * k
    timer = GEMM_TIMER
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _lib.load().cfhip_gemm_bf16(
        a.data_ptr(), b.data_ptr(), out.data_ptr(), _p(bias), _p(aux_in), _p(aux_out), m, n, k, lda,
        ldb, ldc, int(a_trans), int(b_trans), epilogue, 1 if odt is f32 else 0,
        int(accumulate), split_k, _p(ws), ws_bytes, _p(bias_grad), int(bias_grad_accumulate), _stream(),
    )
    _lib.check(rc, "gemm")
    if timer is not None:
        e1.record()
        layout = "tn" if a_trans else ("nn" if b_trans else "nt")
        timer.records.append(((layout, m, n, k, int(epilogue)), e0, e1))
    return out


def gemm_grouped_tn(problems: list) -> None:
    """Weight gradients of several Linear layers in ONE launch (cfhip_gemm_bf16_grouped_tn).

    `problems`: list of (dy [K, M] bf16, x [K, N] bf16, out [M, N] f32, accumulate, bias_grad f32 [M] or None,
    bias_grad_accumulate): out (+)= dy^T x, bias_grad (+)= colsum(dy).  256 x 256 tiles of all problems share the chip and
    every tile runs its whole reduction: no split-K workspace, no reduce pass."""
    if not problems:
        return
    arr = (_lib.GemmProblem * len(problems))()
    flops = 0.0
    for i, (dy, x, out, acc, bg, bg_acc) in enumerate(problems):
        _need(dy, bf16, "dy")
        _need(x, bf16, "x")
        _need(out, f32, "out")
        k, m, lda = _mat(dy, "dy")
        kb, n, ldb = _mat(x, "x")
        if k != kb or tuple(out.shape) != (m, n) or out.stride(1) != 1:
            raise ValueError(f"cfhip gemm_grouped_tn: problem {i}: dy {tuple(dy.shape)} x {tuple(x.shape)} out {tuple(out.shape)}")
        if bg is not None:
            _need(bg, f32, "bias_grad")
            if bg.numel() != m or not bg.is_contiguous():
                raise ValueError("cfhip gemm_grouped_tn: bias_grad must be a contiguous f32 [M]")
        pr = arr[i]
        pr.A, pr.B, pr.C, pr.bias_grad = dy.data_ptr(), x.data_ptr(), out.data_ptr(), _p(bg)
        pr.M, pr.N, pr.K = m, n, k
        pr.lda, pr.ldb, pr.ldc = lda, ldb, out.stride(0)
        pr.accumulate, pr.bias_grad_accumulate = int(bool(acc)), int(bool(bg_acc))
        flops += 2.0 * m * n * k
    if FLOP_COUNTER is not None:
        FLOP_COUNTER.flops["gemm"] += flops
    timer = GEMM_TIMER
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if _lib.RECORDER is not None:
        # the operand addresses of this launch live in a host table, not in the argument tuple: a launch plan that moves its
        # input (fused.StackPlan.repoint) must be able to find and rewrite them — kind-3 entry = the table
        _lib.RECORDER.append((3, arr, len(problems)))
    rc = _lib.load().cfhip_gemm_bf16_grouped_tn(ctypes.cast(arr, ctypes.c_void_p), len(problems), _stream())
    _lib.check(rc, "gemm_grouped_tn")
    if timer is not None:
        e1.record()
        timer.records.append((("tn-grouped", tuple((int(pr.M), int(pr.N), int(pr.K)) for pr in arr), 0, 0, 0), e0, e1))


def colsum(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """out[n] (f32) (+)= sum_m x[m, n]   (bias gradient)."""
    _need(x, bf16, "x")
    m, n, ldx = _mat(x, "x")
    if out is None:
        out = torch.empty((n,), dtype=f32, device=x.device)
        accumulate = False
    _need(out, f32, "out")
    lib = _lib.load()
    nbytes = lib.cfhip_colsum_workspace(m, n)
    ws = torch.empty((max(nbytes, 4) // 4,), dtype=f32, device=x.device)
    rc = lib.cfhip_colsum_bf16(x.data_ptr(), out.data_ptr(), m, n, ldx, int(accumulate), ws.data_ptr(),
                               ws.numel() * 4, _stream())
    _lib.check(rc, "colsum")
    return out


# ---------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------


def layernorm_fwd(
    x: Tensor, gamma: Tensor, beta: Tensor, eps: float, out: Optional[Tensor] = None,
    mean: Optional[Tensor] = None, rstd: Optional[Tensor] = None, out_f32: bool = False,
) -> Tuple[Tensor, Tensor, Tensor]:
    """x: bf16 or f32 [M, D] (row stride free) -> y bf16 (f32 with `out_f32`, f32 rows only) [M, D], mean f32 [M], rstd f32 [M]
    (`out` / `mean` / `rstd`: caller-owned destinations, e.g. row slices of larger tensors)."""
    _need(x, x.dtype if x.dtype in (bf16, f32) else bf16, "x")
    _need(gamma, f32, "gamma")
    _need(beta, f32, "beta")
    m, d, xs = _mat(x, "x")
    if out is not None:
        out_f32 = out.dtype == f32
    if out_f32 and x.dtype != f32:
        raise ValueError("cfhip layernorm_fwd: an f32 output is built for f32 rows only")
    y = torch.empty((m, d), dtype=f32 if out_f32 else bf16, device=x.device) if out is None else out
    mean = torch.empty((m,), dtype=f32, device=x.device) if mean is None else mean
    rstd = torch.empty((m,), dtype=f32, device=x.device) if rstd is None else rstd
    rc = _lib.load().cfhip_layernorm_fwd(
        x.data_ptr(), int(x.dtype == f32) | (2 if out_f32 else 0), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
        rstd.data_ptr(), m, d, xs, y.stride(0), float(eps), _stream(),
    )
    _lib.check(rc, "layernorm_fwd")
    return y, mean, rstd


def layernorm4d_fwd(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    """The reference's 4-D `LN` (norms.py:30-46) on a contiguous bf16 [B, C, H, W]: per-sample mean / UNBIASED std over C*H*W,
    y = (x - mean) / (std + eps) * weight[c] + bias[c].  Returns (y bf16, mean f32 [B], std f32 [B])."""
    _need(x, bf16, "x")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError("cfhip layernorm4d_fwd: a contiguous [B, C, H, W] tensor is expected")
    if (weight is None) != (bias is None):
        raise ValueError("cfhip layernorm4d_fwd: weight and bias come together")
    b, c, h, w = x.shape
    lib = _lib.load()
    nbytes = lib.cfhip_layernorm4d_workspace(b, c, h * w)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    y = torch.empty_like(x)
    mean = torch.empty((b,), dtype=f32, device=x.device)
    std = torch.empty((b,), dtype=f32, device=x.device)
    _lib.check(lib.cfhip_layernorm4d_fwd(x.data_ptr(), _p(weight), _p(bias), y.data_ptr(), mean.data_ptr(), std.data_ptr(), b, c, h * w,
                                         float(eps), ws.data_ptr(), nbytes, _stream()), "layernorm4d_fwd")
    return y, mean, std


def layernorm4d_bwd(dy: Tensor, x: Tensor, weight: Optional[Tensor], mean: Tensor, std: Tensor, eps: float, *, want_dx: bool = True,
                    dweight: Optional[Tensor] = None, dbias: Optional[Tensor] = None, accumulate: bool = False) -> Optional[Tensor]:
    """dx bf16 (or None), dweight / dbias f32 [C] (+)= in place when given"""
    _need(dy, bf16, "dy")
    _need(x, bf16, "x")
    if dy.shape != x.shape or not dy.is_contiguous() or not x.is_contiguous():
        raise ValueError("cfhip layernorm4d_bwd: dy and x must be contiguous [B, C, H, W] of one shape")
    b, c, h, w = x.shape
    lib = _lib.load()
    nbytes = lib.cfhip_layernorm4d_workspace(b, c, h * w)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    dx = torch.empty_like(x) if want_dx else None
    _lib.check(lib.cfhip_layernorm4d_bwd(dy.data_ptr(), x.data_ptr(), _p(weight), mean.data_ptr(), std.data_ptr(), _p(dx), _p(dweight),
                                         _p(dbias), int(accumulate), b, c, h * w, float(eps), ws.data_ptr(), nbytes, _stream()),
               "layernorm4d_bwd")
    return dx


def layernorm_bwd(
    dy: Tensor,
    x: Tensor,
    gamma: Tensor,
    mean: Tensor,
    rstd: Tensor,
    *,
    dx_add: Optional[Tensor] = None,
    dgamma: Optional[Tensor] = None,
    dbeta: Optional[Tensor] = None,
    accumulate: bool = False,
    want_dx: bool = True,
    want_param_grads: bool = True,
    dx_out: Optional[Tensor] = None,
    dx_add_lo: Optional[Tensor] = None,
    dx_lo_out: Optional[Tensor] = None,
) -> Tuple[Optional[Tensor], Optional[Tensor], Optional[Tensor]]:
    """x bf16 or f32.  Returns (dx bf16 [M, D] (+ dx_add), dgamma f32 [D], dbeta f32 [D]).  `dx_out`: dense bf16 [M, D]
    destination for dx (e.g. a row slice of a full-batch tensor).  `dx_add_lo` / `dx_lo_out`: the SECOND bf16 words of dx_add / dx
    when the residual-gradient stream is carried in two words (cfhip_layernorm_bwd2), dense [M, D] like the first."""
    _need(dy, bf16, "dy")
    _need(x, x.dtype if x.dtype in (bf16, f32) else bf16, "x")
    m, d, xs = _mat(x, "x")
    _, _, dys = _mat(dy, "dy")
    if dx_out is not None and want_dx:
        _need(dx_out, bf16, "dx_out")
        if tuple(dx_out.shape) != (m, d) or dx_out.stride(0) != d or dx_out.stride(1) != 1:
            raise ValueError("cfhip layernorm_bwd: dx_out must be a dense [M, D] bf16 tensor")
        dx = dx_out
    else:
        dx = torch.empty((m, d), dtype=bf16, device=x.device) if want_dx else None
    if dx_add is not None:
        _need(dx_add, bf16, "dx_add")
        if tuple(dx_add.shape) != (m, d) or dx_add.stride(0) != d or dx_add.stride(1) != 1:
            raise ValueError("cfhip layernorm_bwd: dx_add must be a dense [M, D] bf16 tensor")
    if not want_param_grads:
        dgamma = dbeta = None
    elif dgamma is None:
        dgamma = torch.empty((d,), dtype=f32, device=x.device)
        dbeta = torch.empty((d,), dtype=f32, device=x.device)
        accumulate = False
    lib = _lib.load()
    nbytes = lib.cfhip_layernorm_bwd_workspace(m, d) if want_param_grads else 0
    ws = torch.empty((max(nbytes, 4) // 4,), dtype=f32, device=x.device)
    if want_dx and (dx_add_lo is not None or dx_lo_out is not None):
        _check_lo(dx_add_lo, dx_lo_out, m, d)
        rc = lib.cfhip_layernorm_bwd2(
            dy.data_ptr(), x.data_ptr(), int(x.dtype == f32), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(dx_add),
            _p(dx_add_lo), _p(dx), _p(dx_lo_out), _p(dgamma), _p(dbeta), m, d, dys, xs, d, int(accumulate), ws.data_ptr(), nbytes,
            None, _stream())
        _lib.check(rc, "layernorm_bwd2")
        return dx, dgamma, dbeta
    rc = lib.cfhip_layernorm_bwd(
        dy.data_ptr(), x.data_ptr(), int(x.dtype == f32), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
        _p(dx_add),
        _p(dx), _p(dgamma), _p(dbeta), m, d, dys, xs, d,
        int(accumulate), ws.data_ptr(), nbytes, _stream(),
    )
    _lib.check(rc, "layernorm_bwd")
    return dx, dgamma, dbeta


def _check_lo(dx_add_lo: Optional[Tensor], dx_lo_out: Optional[Tensor], m: int, d: int) -> None:
    for t, name in ((dx_add_lo, "dx_add_lo"), (dx_lo_out, "dx_lo_out")):
        if t is not None:
            _need(t, bf16, name)
            if tuple(t.shape) != (m, d) or t.stride(0) != d or t.stride(1) != 1:
                raise ValueError(f"cfhip layernorm_bwd: {name} must be a dense [M, D] bf16 tensor")


def split_f32(src: Tensor, hi: Optional[Tensor] = None, lo: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """f32 -> (hi, lo) bf16 with hi = bf16(v), lo = bf16(v - hi): the two words of the residual-gradient stream (cfhip_split_f32_bf16x2)"""
    _need(src, f32, "src")
    src = src.contiguous()
    hi = torch.empty(src.shape, dtype=bf16, device=src.device) if hi is None else hi
    lo = torch.empty(src.shape, dtype=bf16, device=src.device) if lo is None else lo
    _lib.check(_lib.load().cfhip_split_f32_bf16x2(src.data_ptr(), hi.data_ptr(), lo.data_ptr(), src.numel(), _stream()), "split_f32_bf16x2")
    return hi, lo


def join_bf16x2(hi: Tensor, lo: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """(hi, lo) bf16 -> f32 hi + lo (exact) (cfhip_join_bf16x2_f32)"""
    _need(hi, bf16, "hi")
    _need(lo, bf16, "lo")
    if not (hi.is_contiguous() and lo.is_contiguous() and hi.shape == lo.shape):
        raise ValueError("cfhip join_bf16x2: two dense bf16 tensors of one shape expected")
    out = torch.empty(hi.shape, dtype=f32, device=hi.device) if out is None else out
    _lib.check(_lib.load().cfhip_join_bf16x2_f32(hi.data_ptr(), lo.data_ptr(), out.data_ptr(), hi.numel(), _stream()), "join_bf16x2_f32")
    return out


def layernorm_bwd_partials(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, *, dx_add: Optional[Tensor] = None,
                           dx_out: Optional[Tensor] = None, dx_add_lo: Optional[Tensor] = None,
                           dx_lo_out: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, int]:
    """The row kernel of `layernorm_bwd` alone: dx (+ dx_add) and the per-workgroup partial sums of dgamma | dbeta.  Returns
    (dx, workspace, rows) for `layernorm_bwd_reduce` — which may run on another stream (cfhip_layernorm_bwd_partials / _reduce)."""
    _need(dy, bf16, "dy")
    m, d, xs = _mat(x, "x")
    _, _, dys = _mat(dy, "dy")
    dx = dx_out if dx_out is not None else torch.empty((m, d), dtype=bf16, device=x.device)
    lib = _lib.load()
    nbytes = lib.cfhip_layernorm_bwd_workspace(m, d)
    ws = torch.empty((max(nbytes, 4) // 4,), dtype=f32, device=x.device)
    rows = ctypes.c_int(0)
    if dx_add_lo is not None or dx_lo_out is not None:
        _check_lo(dx_add_lo, dx_lo_out, m, d)
        rc = lib.cfhip_layernorm_bwd2(dy.data_ptr(), x.data_ptr(), int(x.dtype == f32), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                      _p(dx_add), _p(dx_add_lo), dx.data_ptr(), _p(dx_lo_out), None, None, m, d, dys, xs, dx.stride(0), 0,
                                      ws.data_ptr(), nbytes, ctypes.byref(rows), _stream())
        _lib.check(rc, "layernorm_bwd2 (partials)")
        return dx, ws, rows.value
    rc = lib.cfhip_layernorm_bwd_partials(dy.data_ptr(), x.data_ptr(), int(x.dtype == f32), gamma.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), _p(dx_add), dx.data_ptr(), m, d, dys, xs, dx.stride(0), ws.data_ptr(),
                                          nbytes, ctypes.byref(rows), _stream())
    _lib.check(rc, "layernorm_bwd_partials")
    return dx, ws, rows.value


def layernorm_bwd_reduce(ws: Tensor, rows: int, d: int, dgamma: Tensor, dbeta: Tensor, accumulate: bool) -> None:
    rc = _lib.load().cfhip_layernorm_bwd_reduce(ws.data_ptr(), rows, d, dgamma.data_ptr(), dbeta.data_ptr(), int(accumulate), _stream())
    _lib.check(rc, "layernorm_bwd_reduce")


# ---------------------------------------------------------------------------------------------
# attention (head_dim 64).  q / k / v / o are [B, T, H*64] views (any batch / token strides).
# ---------------------------------------------------------------------------------------------


def _bth(t: Tensor, name: str) -> Tuple[int, int, int, int, int]:
    _need(t, bf16, name)
    if t.dim() != 3 or t.stride(2) != 1:
        raise ValueError(f"cfhip attention: `{name}` must be [B, T, H*head_dim] with a contiguous last dim")
    return t.shape[0], t.shape[1], t.shape[2], t.stride(0), t.stride(1)


def _mask_args(mask: Optional[Tensor], b: int, h: int, tq: int, tk: int):
    """mask: bool/uint8 'keep' mask broadcastable to [B, H, Tq, Tk] (True = attend)."""
    if mask is None:
        return None, None, 0, 0, 0
    if mask.dtype == torch.bool:
        mask = mask.to(torch.uint8)
    if mask.dtype != torch.uint8:
        raise TypeError("cfhip attention: mask must be bool or uint8")
    while mask.dim() < 4:
        mask = mask.unsqueeze(0)
    mask = mask.expand(b, h, tq, tk)
    if mask.stride(3) != 1:
        mask = mask.contiguous()
    return mask, mask.data_ptr(), mask.stride(0), mask.stride(1), mask.stride(2)


def attn_fwd(
    q: Tensor, k: Tensor, v: Tensor, num_heads: int, *, mask: Optional[Tensor] = None,
    causal: bool = False, scale: Optional[float] = None, head_dim: int = 64, dropout_p: float = 0.0,
    seed: int = 0, offset: int = 0, out: Optional[Tensor] = None, lse: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor]:
    """Returns (o bf16 [B, Tq, H*head_dim] contiguous, lse f32 [B, H, Tq]).  head_dim 64 with both lengths <= 256
    takes the LDS-resident kernels; anything else (head_dim any multiple of 8 up to 192) the chunked general ones.
    `dropout_p` > 0: dropout on the attention probabilities, mask = f(seed, offset) (see `attn_dropout_blocks`)."""
    b, tq, d, q_sb, q_st = _bth(q, "q")
    _, tk, _, k_sb, k_st = _bth(k, "k")
    _, _, _, v_sb, v_st = _bth(v, "v")
    if d != num_heads * head_dim or head_dim % 8 or not 8 <= head_dim <= 192:
        raise ValueError(f"cfhip attention: embed {d} != heads {num_heads} x head_dim {head_dim} "
                         "(head_dim: a multiple of 8 in [8, 192])")
    if (k_sb, k_st) != (v_sb, v_st):
        raise ValueError("cfhip attention: k and v must share batch / token strides")
    if scale is None:
        scale = 1.0 / math.sqrt(float(head_dim))
    o = torch.empty((b, tq, d), dtype=bf16, device=q.device) if out is None else out
    lse = torch.empty((b, num_heads, tq), dtype=f32, device=q.device) if lse is None else lse
    keep, mp, ms_b, ms_h, ms_q = _mask_args(mask, b, num_heads, tq, tk)
    if FLOP_COUNTER is not None:
        FLOP_COUNTER.flops["attention"] += 4.0 * b * num_heads * tq * tk * head_dim
    if dropout_p > 0.0:
        rc = _lib.load().cfhip_attn_fwd_dropout(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), mp, b, num_heads, tq,
            tk, int(head_dim), q_sb, q_st, k_sb, k_st, o.stride(0), o.stride(1), ms_b, ms_h, ms_q, float(scale),
            int(causal), float(dropout_p), int(seed), int(offset), _stream(),
        )
    else:
        rc = _lib.load().cfhip_attn_fwd_dh(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), mp, b, num_heads, tq,
            tk, int(head_dim), q_sb, q_st, k_sb, k_st, o.stride(0), o.stride(1), ms_b, ms_h, ms_q, float(scale),
            int(causal), _stream(),
        )
    _lib.check(rc, "attn_fwd")
    return o, lse


def attn_probs(q: Tensor, k: Tensor, lse: Tensor, num_heads: int, *, mask: Optional[Tensor] = None, causal: bool = False,
               scale: Optional[float] = None, head_dim: int = 64) -> Tensor:
    """f32 [B, H, Tq, Tk]: the attention weights behind `attn_fwd`'s output, exp(scale q.k - lse) with masked slots 0
    (cfhip_attn_probs: the reference's `require_weights` path, attentions.py:256-268)."""
    b, tq, d, q_sb, q_st = _bth(q, "q")
    _, tk, _, k_sb, k_st = _bth(k, "k")
    _need(lse, f32, "lse")
    if scale is None:
        scale = 1.0 / math.sqrt(float(head_dim))
    out = torch.empty((b, num_heads, tq, tk), dtype=f32, device=q.device)
    keep, mp, ms_b, ms_h, ms_q = _mask_args(mask, b, num_heads, tq, tk)
    rc = _lib.load().cfhip_attn_probs(q.data_ptr(), k.data_ptr(), lse.data_ptr(), mp, out.data_ptr(), b, num_heads, tq, tk,
                                      int(head_dim), q_sb, q_st, k_sb, k_st, ms_b, ms_h, ms_q, float(scale), int(causal), _stream())
    _lib.check(rc, "attn_probs")
    return out


def attn_probs_bwd(q: Tensor, k: Tensor, lse: Tensor, d_probs: Tensor, num_heads: int, *, mask: Optional[Tensor] = None,
                   causal: bool = False, scale: Optional[float] = None, head_dim: int = 64) -> Tuple[Tensor, Tensor]:
    """(dq, dk) bf16 [B, T, H*head_dim]: what a gradient on the returned weights adds to the input gradients"""
    b, tq, d, q_sb, q_st = _bth(q, "q")
    _, tk, _, k_sb, k_st = _bth(k, "k")
    _need(lse, f32, "lse")
    _need(d_probs, f32, "d_probs")
    if tuple(d_probs.shape) != (b, num_heads, tq, tk) or not d_probs.is_contiguous():
        raise ValueError("cfhip attn_probs_bwd: d_probs must be a contiguous f32 [B, H, Tq, Tk]")
    if scale is None:
        scale = 1.0 / math.sqrt(float(head_dim))
    ws = torch.empty_like(d_probs)
    dq = torch.empty((b, tq, d), dtype=bf16, device=q.device)
    dk = torch.empty((b, tk, d), dtype=bf16, device=q.device)
    keep, mp, ms_b, ms_h, ms_q = _mask_args(mask, b, num_heads, tq, tk)
    rc = _lib.load().cfhip_attn_probs_bwd(q.data_ptr(), k.data_ptr(), lse.data_ptr(), mp, d_probs.data_ptr(), ws.data_ptr(),
                                          dq.data_ptr(), dk.data_ptr(), b, num_heads, tq, tk, int(head_dim), q_sb, q_st, k_sb, k_st,
                                          ms_b, ms_h, ms_q, float(scale), int(causal), _stream())
    _lib.check(rc, "attn_probs_bwd")
    return dq, dk


def attn_dropout_blocks(b: int, num_heads: int, tq: int, tk: int) -> int:
    """Philox counters one attention call with dropout consumes (`PhiloxState.take` this many)"""
    return b * num_heads * ((tq + 3) // 4) * ((tk + 3) // 4)


def attn_dropout_p(dropout_p: float) -> float:
    """the probability the kernels apply: p quantised to 1 / 256 (csrc/attn.hip set_dropout: thresh = lrintf(256 p), clamped to 1 .. 255)"""
    return min(max(int(round(float(dropout_p) * 256.0)), 1), 255) / 256.0


def attn_dropout_mask(b: int, num_heads: int, tq: int, tk: int, dropout_p: float, seed: int, offset: int,
                      device: object = "cuda") -> Tensor:
    """uint8 [B, H, Tq, Tk] (1 = keep): the mask `attn_fwd(..., dropout_p, seed, offset)` applies"""
    out = torch.empty((b, num_heads, tq, tk), dtype=torch.uint8, device=device)
    rc = _lib.load().cfhip_attn_dropout_mask(out.data_ptr(), b, num_heads, tq, tk, float(dropout_p), int(seed),
                                             int(offset), _stream())
    _lib.check(rc, "attn_dropout_mask")
    return out


def attn_bwd(
    q: Tensor, k: Tensor, v: Tensor, o: Tensor, d_o: Tensor, lse: Tensor, num_heads: int, *,
    dq: Tensor, dk: Tensor, dv: Tensor, mask: Optional[Tensor] = None, causal: bool = False,
    scale: Optional[float] = None, parts: int = 3, delta: Optional[Tensor] = None, head_dim: int = 64,
    dropout_p: float = 0.0, seed: int = 0, offset: int = 0,
) -> None:
    """Writes dq / dk / dv (bf16, SAME strides as q / k / v — e.g. views of one packed buffer)."""
    b, tq, d, q_sb, q_st = _bth(q, "q")
    _, tk, _, k_sb, k_st = _bth(k, "k")
    _bth(v, "v")
    _, _, _, o_sb, o_st = _bth(o, "o")
    _, _, _, do_sb, do_st = _bth(d_o, "d_o")
    if (o_sb, o_st) != (do_sb, do_st):
        raise ValueError("cfhip attention: o and d_o must share strides")
    for t, ref, nm in ((dq, q, "dq"), (dk, k, "dk"), (dv, v, "dv")):
        _need(t, bf16, nm)
        if t.shape != ref.shape or t.stride() != ref.stride():
            raise ValueError(f"cfhip attention: `{nm}` must have the shape and strides of its primal")
    if scale is None:
        scale = 1.0 / math.sqrt(float(head_dim))
    if delta is None:
        delta = torch.empty((b, num_heads, tq), dtype=f32, device=q.device)
    keep, mp, ms_b, ms_h, ms_q = _mask_args(mask, b, num_heads, tq, tk)
    if FLOP_COUNTER is not None:
        FLOP_COUNTER.flops["attention"] += 10.0 * b * num_heads * tq * tk * head_dim
    if dropout_p > 0.0:
        rc = _lib.load().cfhip_attn_bwd_dropout(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(),
            delta.data_ptr(), mp, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), b, num_heads, tq, tk, int(head_dim),
            q_sb, q_st, k_sb, k_st, o_sb, o_st, ms_b, ms_h, ms_q, float(scale), int(causal), int(parts),
            float(dropout_p), int(seed), int(offset), _stream(),
        )
    else:
        rc = _lib.load().cfhip_attn_bwd_dh(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(),
            delta.data_ptr(), mp, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), b, num_heads, tq, tk, int(head_dim),
            q_sb, q_st, k_sb, k_st, o_sb, o_st, ms_b, ms_h, ms_q, float(scale), int(causal), int(parts), _stream(),
        )
    _lib.check(rc, "attn_bwd")


# ---------------------------------------------------------------------------------------------
# ViT glue
# ---------------------------------------------------------------------------------------------


def im2row(img: Tensor, patch: int) -> Tensor:
    """img [B, C, H, W] (f32 or bf16, contiguous) -> bf16 [B*gh*gw, C*P*P] in (c, ph, pw) order."""
    if img.dtype not in (f32, bf16) or not img.is_cuda or not img.is_contiguous() or img.dim() != 4:
        raise ValueError("cfhip im2row: img must be a contiguous f32/bf16 [B, C, H, W] device tensor")
    b, c, hh, ww = img.shape
    rows = torch.empty((b * (hh // patch) * (ww // patch), c * patch * patch), dtype=bf16, device=img.device)
    rc = _lib.load().cfhip_im2row(img.data_ptr(), int(img.dtype == bf16), rows.data_ptr(), b, c, hh, ww,
                                  patch, _stream())
    _lib.check(rc, "im2row")
    return rows


def assemble_tokens_fwd(patches: Tensor, head_token: Tensor, pos: Tensor, b: int,
                        out_dtype: torch.dtype = f32) -> Tensor:
    """patches bf16 [B*Np, D]; head_token f32 [D]; pos f32 [(Np+1)*D] -> x0 [B, Np+1, D] (f32 residual
    stream by default: head_token / pos are f32 parameters, the reference's sum type-promotes)."""
    _need(patches, bf16, "patches")
    _need(head_token, f32, "head_token")
    _need(pos, f32, "pos")
    d = patches.shape[-1]
    np_ = patches.numel() // (b * d)
    x0 = torch.empty((b, np_ + 1, d), dtype=out_dtype, device=patches.device)
    rc = _lib.load().cfhip_assemble_tokens_fwd(patches.data_ptr(), head_token.data_ptr(), pos.data_ptr(),
                                               x0.data_ptr(), int(out_dtype == f32), b, np_, d, _stream())
    _lib.check(rc, "assemble_tokens_fwd")
    return x0


def assemble_tokens_bwd(
    dx0: Tensor, dhead: Optional[Tensor], dpos: Optional[Tensor], accumulate: bool,
    want_dpatches: bool = True,
) -> Optional[Tensor]:
    """dx0 bf16 [B, Np+1, D] contiguous -> dpatches bf16 [B*Np, D]; dhead / dpos f32 (+)=."""
    _need(dx0, bf16, "dx0")
    if not dx0.is_contiguous():
        raise ValueError("cfhip assemble_tokens_bwd: dx0 must be contiguous")
    b, t, d = dx0.shape
    dpatches = torch.empty((b * (t - 1), d), dtype=bf16, device=dx0.device) if want_dpatches else None
    rc = _lib.load().cfhip_assemble_tokens_bwd(dx0.data_ptr(), _p(dpatches), _p(dhead), _p(dpos), b, t - 1,
                                               d, int(accumulate), _stream())
    _lib.check(rc, "assemble_tokens_bwd")
    return dpatches


# ---------------------------------------------------------------------------------------------
# element-wise
# ---------------------------------------------------------------------------------------------


def to_bf16(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """f32 -> bf16 (round-to-nearest-even) through the HIP cast kernel; bf16 passes through."""
    if x.dtype == bf16 and out is None:
        return x
    _need(x, f32, "x")
    if not x.is_contiguous():
        x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=bf16, device=x.device)
    rc = _lib.load().cfhip_cast_f32_to_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream())
    _lib.check(rc, "cast_f32_to_bf16")
    return out


def to_f32(x: Tensor) -> Tensor:
    if x.dtype == f32:
        return x
    _need(x, bf16, "x")
    if not x.is_contiguous():
        x = x.contiguous()
    out = torch.empty(x.shape, dtype=f32, device=x.device)
    rc = _lib.load().cfhip_cast_bf16_to_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream())
    _lib.check(rc, "cast_bf16_to_f32")
    return out


def _ew(fn_name: str, *tensors: Tensor) -> Tensor:
    for i, t in enumerate(tensors):
        _need(t, bf16, f"arg{i}")
        if not t.is_contiguous():
            raise ValueError(f"cfhip {fn_name}: tensors must be contiguous")
    out = torch.empty_like(tensors[0])
    fn = getattr(_lib.load(), fn_name)
    rc = fn(*[t.data_ptr() for t in tensors], out.data_ptr(), out.numel(), _stream())
    _lib.check(rc, fn_name)
    return out


def gelu_fwd(x: Tensor) -> Tensor:
    return _ew("cfhip_gelu_fwd", x)


def gelu_bwd(dy: Tensor, x: Tensor) -> Tensor:
    return _ew("cfhip_gelu_bwd", dy, x)


def quick_gelu_fwd(x: Tensor) -> Tensor:
    return _ew("cfhip_quick_gelu_fwd", x)


def quick_gelu_bwd(dy: Tensor, x: Tensor) -> Tensor:
    return _ew("cfhip_quick_gelu_bwd", dy, x)


def add(a: Tensor, b: Tensor) -> Tensor:
    return _ew("cfhip_add_bf16", a, b)


def transpose(x: Tensor) -> Tensor:
    _need(x, bf16, "x")
    r, c, ld = _mat(x, "x")
    out = torch.empty((c, r), dtype=bf16, device=x.device)
    rc = _lib.load().cfhip_transpose_bf16(x.data_ptr(), out.data_ptr(), r, c, ld, out.stride(0), _stream())
    _lib.check(rc, "transpose")
    return out


# ---------------------------------------------------------------------------------------------
# optimiser / loss
# ---------------------------------------------------------------------------------------------


def adam_step(
    p: Tensor, g: Tensor, m: Tensor, v: Tensor, p_bf16: Optional[Tensor], *, lr: float, beta1: float,
    beta2: float, eps: float, weight_decay: float, decoupled: bool, step: int, grad_scale: float = 1.0,
) -> None:
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _need(t, f32, nm)
        if not t.is_contiguous() or t.numel() != p.numel():
            raise ValueError("cfhip adam_step: flat contiguous f32 buffers of equal length expected")
    if p_bf16 is not None:
        _need(p_bf16, bf16, "p_bf16")
    rc = _lib.load().cfhip_adam_step(
        p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(p_bf16), p.numel(), float(lr),
        float(beta1), float(beta2), float(eps), float(weight_decay), int(decoupled), int(step),
        float(grad_scale), _stream(),
    )
    _lib.check(rc, "adam_step")


def sumsq(g: Tensor) -> Tensor:
    _need(g, f32, "g")
    out = torch.zeros((1,), dtype=f32, device=g.device)
    rc = _lib.load().cfhip_sumsq_f32(g.data_ptr(), out.data_ptr(), g.numel(), _stream())
    _lib.check(rc, "sumsq")
    return out


def softmax_xent(logits: Tensor, labels: Tensor, grad_scale: float, want_grad: bool = True):
    """Returns (loss_sum f32 [1], dlogits f32 [B, C] = (softmax - onehot) * grad_scale)."""
    _need(logits, f32, "logits")
    if labels.dtype != torch.int64 or not labels.is_cuda:
        raise TypeError("cfhip softmax_xent: labels must be int64 on the device")
    if not logits.is_contiguous():
        logits = logits.contiguous()
    labels = labels.reshape(-1).contiguous()
    b, c = logits.shape
    loss = torch.zeros((1,), dtype=f32, device=logits.device)
    dlogits = torch.empty_like(logits) if want_grad else None
    rc = _lib.load().cfhip_softmax_xent(logits.data_ptr(), labels.data_ptr(), loss.data_ptr(), _p(dlogits),
                                        b, c, float(grad_scale), _stream())
    _lib.check(rc, "softmax_xent")
    return loss, dlogits


def softmax_focal(logits: Tensor, labels: Tensor, grad_scale: float, *, gamma: float = 2.0, eps: float = 1.0e-6,
                  want_grad: bool = True):
    """Focal loss of reference losses/basic.py:170-206: returns (loss_sum f32 [1], dlogits f32 [B, C])."""
    _need(logits, f32, "logits")
    if labels.dtype != torch.int64 or not labels.is_cuda:
        raise TypeError("cfhip softmax_focal: labels must be int64 on the device")
    if not logits.is_contiguous():
        logits = logits.contiguous()
    labels = labels.reshape(-1).contiguous()
    b, c = logits.shape
    loss = torch.zeros((1,), dtype=f32, device=logits.device)
    dlogits = torch.empty_like(logits) if want_grad else None
    rc = _lib.load().cfhip_softmax_focal(logits.data_ptr(), labels.data_ptr(), loss.data_ptr(), _p(dlogits),
                                         b, c, float(gamma), float(eps), float(grad_scale), _stream())
    _lib.check(rc, "softmax_focal")
    return loss, dlogits


# ---------------------------------------------------------------------------------------------
# conv / batch-norm family (K8 general form, K9, K10)
# ---------------------------------------------------------------------------------------------


def conv_out_hw(h: int, w: int, kh: int, kw: int, stride: int, pad: int, dil: int) -> Tuple[int, int]:
    return ((h + 2 * pad - dil * (kh - 1) - 1) // stride + 1, (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1)


def conv_im2row(x: Tensor, kh: int, kw: int, stride: int, pad: int, dil: int = 1) -> Tensor:
    """x [B, C, H, W] (f32 / bf16, contiguous) -> rows bf16 [B*Ho*Wo, Kp], k = (c, ky, kx), Kp = K rounded up to 8."""
    if x.dtype not in (f32, bf16) or not x.is_cuda or not x.is_contiguous() or x.dim() != 4:
        raise ValueError("cfhip conv_im2row: x must be a contiguous f32/bf16 [B, C, H, W] device tensor")
    b, c, h, w = x.shape
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad, dil)
    kp = (c * kh * kw + 7) // 8 * 8
    rows = torch.empty((b * ho * wo, kp), dtype=bf16, device=x.device)
    rc = _lib.load().cfhip_conv_im2row(x.data_ptr(), int(x.dtype == f32), rows.data_ptr(), b, c, h, w, kh, kw,
                                       stride, pad, dil, kp, _stream())
    _lib.check(rc, "conv_im2row")
    return rows


def conv_row2im(drows: Tensor, shape: Tuple[int, int, int, int], kh: int, kw: int, stride: int, pad: int,
                dil: int = 1) -> Tensor:
    """drows bf16 [B*Ho*Wo, Kp] -> dx bf16 [B, C, H, W] (gather-sum of the overlapping windows)."""
    _need(drows, bf16, "drows")
    if not drows.is_contiguous():
        drows = drows.contiguous()
    b, c, h, w = shape
    dx = torch.empty(shape, dtype=bf16, device=drows.device)
    rc = _lib.load().cfhip_conv_row2im(drows.data_ptr(), dx.data_ptr(), b, c, h, w, kh, kw, stride, pad, dil,
                                       drows.shape[1], _stream())
    _lib.check(rc, "conv_row2im")
    return dx


def transpose_batched(x: Tensor) -> Tensor:
    """x [batch, R, C] (f32 / bf16, contiguous) -> bf16 [batch, C, R]."""
    if x.dtype not in (f32, bf16) or not x.is_cuda or not x.is_contiguous() or x.dim() != 3:
        raise ValueError("cfhip transpose_batched: x must be a contiguous f32/bf16 [batch, R, C] device tensor")
    n, r, c = x.shape
    out = torch.empty((n, c, r), dtype=bf16, device=x.device)
    rc = _lib.load().cfhip_transpose_batched(x.data_ptr(), int(x.dtype == f32), out.data_ptr(), n, r, c, _stream())
    _lib.check(rc, "transpose_batched")
    return out


def batchnorm_fwd(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], running_mean: Optional[Tensor],
                  running_var: Optional[Tensor], eps: float, momentum: float, training: bool):
    """x [B, C, *] (f32 / bf16, contiguous) -> (y bf16, mean f32 [C], rstd f32 [C]); running stats updated in place."""
    if x.dtype not in (f32, bf16) or not x.is_cuda or not x.is_contiguous() or x.dim() < 2:
        raise ValueError("cfhip batchnorm_fwd: x must be a contiguous f32/bf16 [B, C, ...] device tensor")
    b, c = x.shape[0], x.shape[1]
    inner = x.numel() // (b * c)
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        if t is not None:
            _need(t, f32, nm)
    y = torch.empty(x.shape, dtype=bf16, device=x.device)
    mean = torch.empty((c,), dtype=f32, device=x.device)
    rstd = torch.empty((c,), dtype=f32, device=x.device)
    rc = _lib.load().cfhip_batchnorm_fwd(x.data_ptr(), int(x.dtype == f32), _p(gamma), _p(beta), y.data_ptr(),
                                         mean.data_ptr(), rstd.data_ptr(), _p(running_mean), _p(running_var), b, c,
                                         inner, float(eps), float(momentum), int(training), _stream())
    _lib.check(rc, "batchnorm_fwd")
    return y, mean, rstd


def batchnorm_bwd(dy: Tensor, x: Tensor, gamma: Optional[Tensor], mean: Tensor, rstd: Tensor, *, training: bool,
                  want_dx: bool = True, dgamma: Optional[Tensor] = None, dbeta: Optional[Tensor] = None,
                  accumulate: bool = False):
    """Returns (dx bf16 | None, dgamma f32 [C], dbeta f32 [C])."""
    _need(dy, bf16, "dy")
    if not dy.is_contiguous():
        dy = dy.contiguous()
    b, c = x.shape[0], x.shape[1]
    inner = x.numel() // (b * c)
    dx = torch.empty(x.shape, dtype=bf16, device=x.device) if want_dx else None
    if dgamma is None:
        dgamma, accumulate = torch.empty((c,), dtype=f32, device=x.device), False
        dbeta = torch.empty((c,), dtype=f32, device=x.device)
    rc = _lib.load().cfhip_batchnorm_bwd(dy.data_ptr(), x.data_ptr(), int(x.dtype == f32), _p(gamma), mean.data_ptr(),
                                         rstd.data_ptr(), _p(dx), dgamma.data_ptr(), dbeta.data_ptr(), b, c, inner,
                                         int(accumulate), int(training), _stream())
    _lib.check(rc, "batchnorm_bwd")
    return dx, dgamma, dbeta


def leaky_relu_fwd(x: Tensor, slope: float) -> Tensor:
    _need(x, bf16, "x")
    if not x.is_contiguous():
        x = x.contiguous()
    y = torch.empty_like(x)
    _lib.check(_lib.load().cfhip_leaky_relu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), float(slope), _stream()),
               "leaky_relu_fwd")
    return y


def leaky_relu_bwd(dy: Tensor, x: Tensor, slope: float) -> Tensor:
    _need(dy, bf16, "dy")
    _need(x, bf16, "x")
    if not dy.is_contiguous():
        dy = dy.contiguous()
    dx = torch.empty_like(x)
    _lib.check(_lib.load().cfhip_leaky_relu_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), float(slope),
                                                _stream()), "leaky_relu_bwd")
    return dx


def avgpool_fwd(x: Tensor) -> Tensor:
    """x bf16 [B, C, H, W] -> bf16 [B, C] (AdaptiveAvgPool2d((1, 1)) + squeeze)."""
    _need(x, bf16, "x")
    if not x.is_contiguous():
        x = x.contiguous()
    b, c = x.shape[0], x.shape[1]
    inner = x.numel() // (b * c)
    y = torch.empty((b, c), dtype=bf16, device=x.device)
    _lib.check(_lib.load().cfhip_avgpool_fwd(x.data_ptr(), y.data_ptr(), b * c, inner, _stream()), "avgpool_fwd")
    return y


def avgpool_bwd(dy: Tensor, shape: Tuple[int, ...]) -> Tensor:
    _need(dy, bf16, "dy")
    if not dy.is_contiguous():
        dy = dy.contiguous()
    b, c = shape[0], shape[1]
    dx = torch.empty(shape, dtype=bf16, device=dy.device)
    inner = dx.numel() // (b * c)
    _lib.check(_lib.load().cfhip_avgpool_bwd(dy.data_ptr(), dx.data_ptr(), b * c, inner, _stream()), "avgpool_bwd")
    return dx


# ---------------------------------------------------------------------------------------------
# row gather / scatter-add, L2 normalisation (CLIP text tower)
# ---------------------------------------------------------------------------------------------


def embedding_fwd(table: Tensor, indices: Tensor, pos: Optional[Tensor] = None, *, period: int = 0,
                  out_dtype: torch.dtype = f32) -> Tensor:
    """out[n] = table[indices[n]] (+ pos[n % period]); table f32 [V, D], indices int64 [...], pos f32 [>= period, D]."""
    _need(table, f32, "table")
    if indices.dtype != torch.int64 or not indices.is_cuda:
        raise TypeError("cfhip embedding_fwd: indices must be int64 on the device")
    if table.dim() != 2 or not table.is_contiguous():
        raise ValueError("cfhip embedding_fwd: table must be a contiguous [V, D] matrix")
    idx = indices.reshape(-1).contiguous()
    v, d = table.shape
    if pos is not None:
        _need(pos, f32, "pos")
        if not pos.is_contiguous() or pos.shape[-1] != d or pos.numel() < period * d:
            raise ValueError("cfhip embedding_fwd: pos must be contiguous [>= period, D]")
    out = torch.empty((idx.numel(), d), dtype=out_dtype, device=table.device)
    rc = _lib.load().cfhip_embedding_fwd(table.data_ptr(), idx.data_ptr(), _p(pos), out.data_ptr(),
                                         int(out_dtype == f32), idx.numel(), d, int(period), v, _stream())
    _lib.check(rc, "embedding_fwd")
    return out.view(*indices.shape, d)


def embedding_bwd(dy: Tensor, indices: Tensor, dtable: Tensor, padding_idx: int = -1) -> None:
    """dtable[indices[n]] += dy[n] (dtable f32 [V, D], owned / zeroed by the caller)."""
    _need(dtable, f32, "dtable")
    if dy.dtype not in (f32, bf16) or not dy.is_cuda:
        raise TypeError("cfhip embedding_bwd: dy must be f32 / bf16 on the device")
    v, d = dtable.shape
    dy2 = dy.reshape(-1, d).contiguous()
    idx = indices.reshape(-1).contiguous()
    rc = _lib.load().cfhip_embedding_bwd(dy2.data_ptr(), int(dy2.dtype == f32), idx.data_ptr(), dtable.data_ptr(),
                                         idx.numel(), d, v, int(padding_idx), _stream())
    _lib.check(rc, "embedding_bwd")


def l2norm_fwd(x: Tensor) -> Tuple[Tensor, Tensor]:
    _need(x, f32, "x")
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    y = torch.empty_like(x2)
    inv = torch.empty((x2.shape[0],), dtype=f32, device=x.device)
    _lib.check(_lib.load().cfhip_l2norm_fwd(x2.data_ptr(), y.data_ptr(), inv.data_ptr(), x2.shape[0], x2.shape[1],
                                            _stream()), "l2norm_fwd")
    return y.view(x.shape), inv


def l2norm_bwd(dy: Tensor, y: Tensor, inv: Tensor) -> Tensor:
    _need(dy, f32, "dy")
    dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
    y2 = y.reshape(-1, y.shape[-1]).contiguous()
    dx = torch.empty_like(dy2)
    _lib.check(_lib.load().cfhip_l2norm_bwd(dy2.data_ptr(), y2.data_ptr(), inv.data_ptr(), dx.data_ptr(), dy2.shape[0],
                                            dy2.shape[1], _stream()), "l2norm_bwd")
    return dx.view(dy.shape)


# ---------------------------------------------------------------------------------------------
# UNet residual-block pieces: GroupNorm (+ additive term, + SiLU), SiLU, x2 resampling, timestep embedding
# ---------------------------------------------------------------------------------------------


def colreduce_f32(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """out[d] (+)= sum_r x[r][d] for a contiguous f32 [R, D] matrix."""
    _need(x, f32, "x")
    if x.dim() != 2 or not x.is_contiguous():
        raise ValueError("cfhip colreduce_f32: contiguous [R, D] matrix expected")
    if out is None:
        out, accumulate = torch.empty((x.shape[1],), dtype=f32, device=x.device), False
    _lib.check(_lib.load().cfhip_colreduce_f32(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], int(accumulate),
                                               _stream()), "colreduce_f32")
    return out


def colreduce2_f32(xa: Tensor, xb: Tensor, out_a: Tensor, out_b: Tensor, accumulate: bool) -> None:
    """out_a[d] (+)= sum_r xa[r][d] and the same for (xb, out_b) — two contiguous f32 [R, D] matrices of one shape — in one launch"""
    _need(xa, f32, "xa")
    _need(xb, f32, "xb")
    if xa.dim() != 2 or xa.shape != xb.shape or not xa.is_contiguous() or not xb.is_contiguous():
        raise ValueError("cfhip colreduce2_f32: two contiguous [R, D] matrices of one shape expected")
    _lib.check(_lib.load().cfhip_colreduce2_f32(xa.data_ptr(), out_a.data_ptr(), xb.data_ptr(), out_b.data_ptr(), xa.shape[0], xa.shape[1],
                                                int(accumulate), _stream()), "colreduce2_f32")


def _gn_affine_stride(gamma: Tensor, beta: Tensor, b: int, c: int) -> int:
    if gamma.shape != beta.shape or not gamma.is_contiguous() or not beta.is_contiguous():
        raise ValueError("cfhip groupnorm: gamma / beta must be contiguous and of one shape")
    if tuple(gamma.shape) == (c,):
        return 0
    if tuple(gamma.shape) == (b, c):
        return c
    raise ValueError(f"cfhip groupnorm: gamma must be [C] or [B, C], got {tuple(gamma.shape)}")


GN_TARGET_WORKGROUPS = 1024  # four 4-wave workgroups per CU; 0: never split
GN_MIN_SLICE = 2048  # elements of `inner` per slice and channel below which splitting stops paying


def gn_splits(b: int, c: int, groups: int, inner: int) -> int:
    """Slices per (sample, group) of the split GroupNorm kernels: enough workgroups to fill the chip when B * G alone does
    not (batch 1 at 256^2: 32 workgroups on 256 CUs), never slices shorter than GN_MIN_SLICE elements per channel."""
    if GN_TARGET_WORKGROUPS <= 0 or inner % 8 != 0:
        return 1
    want = -(-GN_TARGET_WORKGROUPS // (b * groups))
    return max(1, min(64, want, inner // GN_MIN_SLICE))


GN_NHWC_TARGET_WORKGROUPS = 1024
GN_NHWC_MIN_ROWS = 16


GN_NHWC_GROUP_MAX_ROWS = 48  # rows per thread up to which the group form is used
GN_NHWC_GROUP_MIN_WORKGROUPS = 128  # B * G from which one workgroup per (sample, group) fills the chip


def gn_nhwc_splits(b: int, inner: int, c: int = 0, groups: int = 0, backward: bool = False) -> int:
    """0: the group form (one workgroup per (sample, group), ONE launch each way: enough samples, an even number of channels per group);
    otherwise the row slices per sample of the slice form (few samples, e.g. 256^2 x 1): ~4 workgroups per CU, >= GN_NHWC_MIN_ROWS rows each"""
    if groups and c % groups == 0:
        cpg = c // groups
        if b * groups >= GN_NHWC_GROUP_MIN_WORKGROUPS and cpg % 2 == 0 and cpg <= 256:
            # a thread of the group form owns ceil(inner / row lanes) rows as 2 cpg-byte chunks (20-60 bytes of every 640-3 840-byte
            # row on the UNet's upper levels): fine while there are few, but with many rows per thread the slice form's whole-row
            # 16-byte accesses win although it is three launches — alone only beyond the 88 rows the group form can keep in
            # registers (64^2 x 8 x 960: 281 / 246 us forward / backward against 94 / 126, `tools/gn_nhwc_bench.py`), inside the
            # step, beside the weight-gradient GEMMs of the other queue, already from ~50 (64^2 x 8 step: threshold none / 88 / 48 /
            # 32 / 16 / 0 = 59.3 / 58.8 / 58.3 / +0.4 / +1.4 / +3.0 ms, `profiles/r05/gn_nhwc_form_rule_ab.txt`)
            d = cpg // 2  # dwords per row chunk; a thread reads 4 / 2 / 1 of them with one access (conv.hip: gn_group_vw)
            vw = 4 if d % 4 == 0 else 2 if d % 2 == 0 else 1
            row_lanes = 256 // (d // vw)
            if -(-inner // row_lanes) <= GN_NHWC_GROUP_MAX_ROWS:
                return 0
            # (forward: 32 slices — its statistics kernel reads the slice twice, the second time out of L2; backward: 64)
            return max(1, min(64 if backward else 32, inner // 32))
    want = -(-GN_NHWC_TARGET_WORKGROUPS // b)
    return max(1, min(want, inner // GN_NHWC_MIN_ROWS, 4096))


def groupnorm_nhwc_fwd(rows: Tensor, b: int, gamma: Tensor, beta: Tensor, groups: int, eps: float, *, add: Optional[Tensor] = None,
                       silu: bool = False):
    """rows bf16 [B * inner, C] (NHWC) -> (y rows bf16, mean f32 [B * G], rstd f32 [B * G]); y = [SiLU](GN(x + add[b, c])).
    gamma / beta: f32 [C], or [B, C] = one affine per sample."""
    _need(rows, bf16, "rows")
    _need(gamma, f32, "gamma")
    _need(beta, f32, "beta")
    if rows.dim() != 2 or not rows.is_contiguous() or rows.shape[0] % b:
        raise ValueError("cfhip groupnorm_nhwc_fwd: contiguous bf16 [B * inner, C] rows expected")
    c = rows.shape[1]
    inner = rows.shape[0] // b
    affine_bs = _gn_affine_stride(gamma, beta, b, c)
    if add is not None:
        _need(add, f32, "add")
        if tuple(add.shape) != (b, c) or not add.is_contiguous():
            raise ValueError("cfhip groupnorm_nhwc_fwd: add must be contiguous f32 [B, C]")
    lib = _lib.load()
    splits = gn_nhwc_splits(b, inner, c, groups)
    y = torch.empty_like(rows)
    mean = torch.empty((b * groups,), dtype=f32, device=rows.device)
    rstd = torch.empty((b * groups,), dtype=f32, device=rows.device)
    ws = None if splits == 0 else torch.empty((max(4, lib.cfhip_groupnorm_nhwc_workspace(b, c, groups, splits, 0, 0)) // 4,), dtype=f32,
                                              device=rows.device)
    rc = lib.cfhip_groupnorm_nhwc_fwd(rows.data_ptr(), _p(add), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                      rstd.data_ptr(), b, c, groups, inner, float(eps), int(silu), affine_bs, splits, _p(ws), _stream())
    _lib.check(rc, "groupnorm_nhwc_fwd")
    return y, mean, rstd


def groupnorm_nhwc_bwd(dy: Tensor, rows: Tensor, b: int, gamma: Tensor, beta: Tensor, mean: Tensor, rstd: Tensor, groups: int, *,
                       add: Optional[Tensor] = None, silu: bool = False):
    """Returns (dx rows bf16, dgamma_part f32 [B, C], dbeta_part f32 [B, C], dadd f32 [B, C] | None): the per-sample partial sums are
    reduced over B by the caller (`colreduce_f32(part, out=param.grad)`), or ARE the gradient of a per-sample affine."""
    _need(dy, bf16, "dy")
    _need(rows, bf16, "rows")
    if dy.shape != rows.shape or not dy.is_contiguous() or not rows.is_contiguous():
        raise ValueError("cfhip groupnorm_nhwc_bwd: dy and x must be contiguous bf16 rows of one shape")
    c = rows.shape[1]
    inner = rows.shape[0] // b
    affine_bs = _gn_affine_stride(gamma, beta, b, c)
    lib = _lib.load()
    splits = gn_nhwc_splits(b, inner, c, groups, True)
    dx = torch.empty_like(rows)
    dg = torch.empty((b, c), dtype=f32, device=rows.device)
    db = torch.empty((b, c), dtype=f32, device=rows.device)
    dadd = torch.empty((b, c), dtype=f32, device=rows.device) if add is not None else None
    ws = None if splits == 0 else torch.empty((max(4, lib.cfhip_groupnorm_nhwc_workspace(b, c, groups, splits, 1, int(add is not None))) // 4,),
                                              dtype=f32, device=rows.device)
    rc = lib.cfhip_groupnorm_nhwc_bwd(dy.data_ptr(), rows.data_ptr(), _p(add), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                      rstd.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), _p(dadd), b, c, groups, inner,
                                      int(silu), affine_bs, splits, _p(ws), _stream())
    _lib.check(rc, "groupnorm_nhwc_bwd")
    return dx, dg, db, dadd


def upsample2_nhwc(rows: Tensor, b: int, h: int, w: int, backward: bool = False) -> Tensor:
    """nearest x2 on NHWC rows [B * H * W, C] (h, w = the SMALL size): forward -> [B * 4 H W, C]; backward: dy rows [B * 4 H W, C] -> dx"""
    _need(rows, bf16, "rows")
    c = rows.shape[1]
    n_in = b * h * w * (4 if backward else 1)
    if rows.dim() != 2 or not rows.is_contiguous() or rows.shape[0] != n_in:
        raise ValueError("cfhip upsample2_nhwc: contiguous bf16 rows of the stated geometry expected")
    out = torch.empty((b * h * w * (1 if backward else 4), c), dtype=bf16, device=rows.device)
    fn = _lib.load().cfhip_upsample2_nhwc_bwd if backward else _lib.load().cfhip_upsample2_nhwc_fwd
    _lib.check(fn(rows.data_ptr(), out.data_ptr(), b, h, w, c, _stream()), "upsample2_nhwc")
    return out


def groupnorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, *, add: Optional[Tensor] = None,
                  silu: bool = False):
    """x [B, C, ...] f32 / bf16 -> (y bf16, mean f32 [B*G], rstd f32 [B*G]); y = [SiLU](GN(x + add[b, c])).
    gamma / beta: f32 [C], or [B, C] = one affine per sample (the scale-shift norm of residual.py:236-239)."""
    if x.dtype not in (f32, bf16) or not x.is_cuda or not x.is_contiguous() or x.dim() < 2:
        raise ValueError("cfhip groupnorm_fwd: x must be a contiguous f32/bf16 [B, C, ...] device tensor")
    _need(gamma, f32, "gamma")
    _need(beta, f32, "beta")
    b, c = x.shape[0], x.shape[1]
    inner = x.numel() // (b * c)
    affine_bs = _gn_affine_stride(gamma, beta, b, c)
    if add is not None:
        _need(add, f32, "add")
        if tuple(add.shape) != (b, c) or not add.is_contiguous():
            raise ValueError("cfhip groupnorm_fwd: add must be contiguous f32 [B, C]")
    y = torch.empty(x.shape, dtype=bf16, device=x.device)
    mean = torch.empty((b * groups,), dtype=f32, device=x.device)
    rstd = torch.empty((b * groups,), dtype=f32, device=x.device)
    splits = gn_splits(b, c, groups, inner)
    if splits > 1 and x.data_ptr() % 16 == 0:
        ws = torch.empty((b * groups * splits * 2,), dtype=f32, device=x.device)
        rc = _lib.load().cfhip_groupnorm_split_fwd(x.data_ptr(), int(x.dtype == f32), _p(add), gamma.data_ptr(), beta.data_ptr(),
                                                   y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), b, c, groups, inner, float(eps),
                                                   int(silu), affine_bs, splits, ws.data_ptr(), _stream())
        _lib.check(rc, "groupnorm_split_fwd")
        return y, mean, rstd
    rc = _lib.load().cfhip_groupnorm_affine_fwd(x.data_ptr(), int(x.dtype == f32), _p(add), gamma.data_ptr(),
                                                beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), b, c,
                                                groups, inner, float(eps), int(silu), affine_bs, _stream())
    _lib.check(rc, "groupnorm_fwd")
    return y, mean, rstd


def groupnorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, beta: Tensor, mean: Tensor, rstd: Tensor, groups: int, *,
                  add: Optional[Tensor] = None, silu: bool = False, reduce: bool = True):
    """Returns (dx bf16, dgamma, dbeta, dadd f32 [B, C] | None); dgamma / dbeta have the shape of gamma ([C], or [B, C]
    for a per-sample affine).  reduce=False: the per-sample partial sums [B, C] are returned as they are (the caller
    reduces them over B where they belong, e.g. `colreduce_f32(part, out=param.grad)`)."""
    _need(dy, bf16, "dy")
    if not dy.is_contiguous():
        dy = dy.contiguous()
    b, c = x.shape[0], x.shape[1]
    inner = x.numel() // (b * c)
    affine_bs = _gn_affine_stride(gamma, beta, b, c)
    dx = torch.empty(x.shape, dtype=bf16, device=x.device)
    dg_part = torch.empty((b, c), dtype=f32, device=x.device)
    db_part = torch.empty((b, c), dtype=f32, device=x.device)
    dadd = torch.empty((b, c), dtype=f32, device=x.device) if add is not None else None
    splits = gn_splits(b, c, groups, inner)
    if splits > 1 and c // groups <= 128 and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0:
        ws = torch.empty((b * groups * splits * 3 * (c // groups),), dtype=f32, device=x.device)
        rc = _lib.load().cfhip_groupnorm_split_bwd(dy.data_ptr(), x.data_ptr(), int(x.dtype == f32), _p(add), gamma.data_ptr(),
                                                   beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                                   dg_part.data_ptr(), db_part.data_ptr(), _p(dadd), b, c, groups, inner,
                                                   int(silu), affine_bs, splits, ws.data_ptr(), _stream())
        _lib.check(rc, "groupnorm_split_bwd")
        if affine_bs or not reduce:
            return dx, dg_part, db_part, dadd
        return dx, colreduce_f32(dg_part), colreduce_f32(db_part), dadd
    rc = _lib.load().cfhip_groupnorm_affine_bwd(dy.data_ptr(), x.data_ptr(), int(x.dtype == f32), _p(add),
                                                gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                dx.data_ptr(), dg_part.data_ptr(), db_part.data_ptr(), _p(dadd), b, c,
                                                groups, inner, int(silu), affine_bs, _stream())
    _lib.check(rc, "groupnorm_bwd")
    if affine_bs or not reduce:
        return dx, dg_part, db_part, dadd
    return dx, colreduce_f32(dg_part), colreduce_f32(db_part), dadd


TIME_PROJ_MAX = 32  # problems per launch of the grouped time-embedding projection (csrc/conv.hip: TP_MAX)


def _time_proj_table(rows) -> "ctypes.Array":
    """4 int64 per problem, a HOST array that lives until the call returns (the pointers travel in the kernel arguments)"""
    flat = [int(v) for row in rows for v in row]
    return (ctypes.c_int64 * len(flat))(*flat)


def time_proj_fwd(emb: Tensor, weights16, biases, t16: Tensor):
    """out_i = Linear_i(SiLU(emb)) for every (weight bf16 [N_i, K], bias f32 [N_i] or None) in ONE launch; returns the list of f32
    [B, N_i] outputs and fills t16 = bf16 SiLU(emb) [B, K] (cfhip_time_proj_fwd; reference residual.py:226-239 per block)"""
    _need(emb, f32, "emb")
    b, k = emb.shape
    outs = [torch.empty((b, w.shape[0]), dtype=f32, device=emb.device) for w in weights16]
    rows = []
    for w, bias, out in zip(weights16, biases, outs):
        _need(w, bf16, "weight")
        if w.shape[1] != k or not w.is_contiguous() or (bias is not None and (bias.dtype != f32 or not bias.is_contiguous())):
            raise ValueError("cfhip time_proj_fwd: contiguous bf16 [N, K] weights and f32 [N] biases expected")
        rows.append((w.data_ptr(), 0 if bias is None else bias.data_ptr(), out.data_ptr(), w.shape[0]))
    table = _time_proj_table(rows)
    _lib.check(_lib.load().cfhip_time_proj_fwd(emb.data_ptr(), b, k, ctypes.addressof(table), len(rows), t16.data_ptr(), _stream()),
               "time_proj_fwd")
    return outs


def time_proj_bwd(emb: Tensor, weights16, dys, dys16) -> Tensor:
    """d_emb = SiLU'(emb) * sum_i bf16(dY_i) W_i in two launches (partial sums per 64 columns, then their sum in a fixed order);
    dys[i] None = no gradient for output i; dys16[i]: bf16 [B, N_i] buffers that receive the rounded dY_i (or None)"""
    b, k = emb.shape
    rows = []
    nblk = 0
    for w, dy, dy16 in zip(weights16, dys, dys16):
        if dy is not None and (dy.dtype != f32 or not dy.is_contiguous() or dy.shape != (b, w.shape[0])):
            raise ValueError("cfhip time_proj_bwd: contiguous f32 [B, N] gradients expected")
        rows.append((w.data_ptr(), 0 if dy is None else dy.data_ptr(), 0 if dy16 is None else dy16.data_ptr(), w.shape[0]))
        nblk += (w.shape[0] + 63) // 64
    table = _time_proj_table(rows)
    partial = torch.empty((nblk, (b + 7) // 8 * 8, k), dtype=f32, device=emb.device)
    d_emb = torch.empty_like(emb)
    _lib.check(_lib.load().cfhip_time_proj_bwd(emb.data_ptr(), b, k, ctypes.addressof(table), len(rows), partial.data_ptr(),
                                               d_emb.data_ptr(), _stream()), "time_proj_bwd")
    return d_emb


def silu_f32_fwd(x: Tensor) -> Tensor:
    _need(x, f32, "x")
    x = x.contiguous()
    y = torch.empty_like(x)
    _lib.check(_lib.load().cfhip_silu_f32_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "silu_f32_fwd")
    return y


def silu_f32_bwd(dy: Tensor, x: Tensor) -> Tensor:
    _need(dy, f32, "dy")
    _need(x, f32, "x")
    dy, x = dy.contiguous(), x.contiguous()
    dx = torch.empty_like(x)
    _lib.check(_lib.load().cfhip_silu_f32_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), _stream()),
               "silu_f32_bwd")
    return dx


def _resample(name: str, x: Tensor, out_hw: Tuple[int, int], small_hw: Tuple[int, int]) -> Tensor:
    _need(x, bf16, "x")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError(f"cfhip {name}: contiguous bf16 [B, C, H, W] expected")
    b, c = x.shape[0], x.shape[1]
    out = torch.empty((b, c, out_hw[0], out_hw[1]), dtype=bf16, device=x.device)
    rc = getattr(_lib.load(), f"cfhip_{name}")(x.data_ptr(), out.data_ptr(), b * c, small_hw[0], small_hw[1], _stream())
    _lib.check(rc, name)
    return out


def reflect_pad2d_fwd(x: Tensor, pads: Tuple[int, int, int, int]) -> Tensor:
    """nn.ReflectionPad2d((left, right, top, bottom)): x [B, C, H, W] bf16 / f32 -> bf16 [B, C, H + t + b, W + l + r]."""
    _need(x, x.dtype if x.dtype in (bf16, f32) else bf16, "x")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError("cfhip reflect_pad2d: contiguous [B, C, H, W] expected")
    pl, pr, pt, pb = (int(v) for v in pads)
    b, c, h, w = x.shape
    out = torch.empty((b, c, h + pt + pb, w + pl + pr), dtype=bf16, device=x.device)
    rc = _lib.load().cfhip_reflect_pad2d_fwd(x.data_ptr(), int(x.dtype == f32), out.data_ptr(), b * c, h, w, pl, pr, pt, pb, _stream())
    _lib.check(rc, "reflect_pad2d_fwd")
    return out


def reflect_pad2d_bwd(dy: Tensor, pads: Tuple[int, int, int, int]) -> Tensor:
    _need(dy, bf16, "dy")
    if dy.dim() != 4 or not dy.is_contiguous():
        raise ValueError("cfhip reflect_pad2d_bwd: contiguous bf16 [B, C, Ho, Wo] expected")
    pl, pr, pt, pb = (int(v) for v in pads)
    b, c, ho, wo = dy.shape
    h, w = ho - pt - pb, wo - pl - pr
    dx = torch.empty((b, c, h, w), dtype=bf16, device=dy.device)
    rc = _lib.load().cfhip_reflect_pad2d_bwd(dy.data_ptr(), dx.data_ptr(), b * c, h, w, pl, pr, pt, pb, _stream())
    _lib.check(rc, "reflect_pad2d_bwd")
    return dx


def upsample2_fwd(x: Tensor) -> Tensor:
    h, w = x.shape[2], x.shape[3]
    return _resample("upsample2_fwd", x, (2 * h, 2 * w), (h, w))


def upsample2_bwd(dy: Tensor) -> Tensor:
    h, w = dy.shape[2] // 2, dy.shape[3] // 2
    return _resample("upsample2_bwd", dy, (h, w), (h, w))


def avgpool2_fwd(x: Tensor) -> Tensor:
    if x.shape[2] % 2 or x.shape[3] % 2:
        raise ValueError("cfhip avgpool2: even spatial sizes expected")
    h, w = x.shape[2] // 2, x.shape[3] // 2
    return _resample("avgpool2_fwd", x, (h, w), (h, w))


def avgpool2_bwd(dy: Tensor) -> Tensor:
    h, w = dy.shape[2], dy.shape[3]
    return _resample("avgpool2_bwd", dy, (2 * h, 2 * w), (h, w))


def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    """t int64 [B] -> f32 [B, dim] = [cos(t f) | sin(t f)] (reference multimodal/diffusion/unet.py:52-74)."""
    if t.dtype != torch.int64 or not t.is_cuda or t.dim() != 1:
        raise TypeError("cfhip timestep_embedding: t must be int64 [B] on the device")
    out = torch.empty((t.shape[0], dim), dtype=f32, device=t.device)
    _lib.check(_lib.load().cfhip_timestep_embedding(t.contiguous().data_ptr(), out.data_ptr(), t.shape[0], dim,
                                                    float(max_period), _stream()), "timestep_embedding")
    return out


def conv3x3_nhwc(x_rows: Tensor, wk: Tensor, bias: Optional[Tensor], b: int, h: int, w: int) -> Tensor:
    """Implicit-GEMM 3x3 / stride 1 / pad 1 convolution: x_rows bf16 [B*H*W, Cin] (NHWC), wk bf16 [Cout, 9*Cin] with
    k = (ky*3 + kx)*Cin + c, bias f32 [Cout] | None -> bf16 [B*H*W, Cout]."""
    _need(x_rows, bf16, "x_rows")
    _need(wk, bf16, "wk")
    m, cin = x_rows.shape
    cout = wk.shape[0]
    if m != b * h * w or wk.shape[1] != 9 * cin or not x_rows.is_contiguous() or not wk.is_contiguous():
        raise ValueError(f"cfhip conv3x3_nhwc: x_rows {tuple(x_rows.shape)} / wk {tuple(wk.shape)} do not match "
                         f"B={b} H={h} W={w} (dense NHWC rows and [Cout, 9*Cin] filters expected)")
    if bias is not None:
        _need(bias, f32, "bias")
    y = torch.empty((m, cout), dtype=bf16, device=x_rows.device)
    if FLOP_COUNTER is not None:
        FLOP_COUNTER.flops["conv3x3"] += 2.0 * m * cout * 9 * cin
    lib = _lib.load()
    nbytes = lib.cfhip_conv3x3_workspace(b, h, w, cin, cout)  # > 0 when the shape splits its reduction
    ws = torch.empty((max(nbytes, 4) // 4,), dtype=f32, device=x_rows.device)
    _lib.check(lib.cfhip_conv3x3_nhwc_bf16(x_rows.data_ptr(), wk.data_ptr(), _p(bias), y.data_ptr(), b, h, w, cin, cout,
                                           ws.data_ptr(), nbytes, _stream()), "conv3x3_nhwc")
    return y


def conv3x3_wgrad_ok(b: int, h: int, w: int) -> bool:
    return h >= 2 and w >= 2 and h < 65536 and w < 65536 and b * h * w * max(h, w) < (1 << 32)


def _gconv_args(x_shape, w: Tensor, stride: int, pad: int, dil: int, groups: int):
    b, cin, h, wd = x_shape
    cout, cgi, kh, kw = w.shape
    if cin % groups or cout % groups or cgi != cin // groups:
        raise ValueError(f"cfhip grouped conv: weight {tuple(w.shape)} does not fit Cin {cin} with groups {groups}")
    return [b, cin, h, wd, cout, kh, kw, int(stride), int(pad), int(dil), int(groups)]


def conv2d_grouped_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], stride: int, pad: int, dil: int, groups: int) -> Tensor:
    """F.conv2d(x, w, bias, stride, pad, dil, groups) for groups > 1 (cfhip_conv2d_grouped_fwd): x bf16 NCHW, w bf16
    [Cout, Cin / groups, kh, kw], bias f32 -> y bf16 NCHW"""
    _need(x, bf16, "x")
    _need(w, bf16, "w")
    args = _gconv_args(x.shape, w, stride, pad, dil, groups)
    ho, wo = conv_out_hw(x.shape[2], x.shape[3], w.shape[2], w.shape[3], stride, pad, dil)
    y = torch.empty((x.shape[0], w.shape[0], ho, wo), dtype=bf16, device=x.device)
    rc = _lib.load().cfhip_conv2d_grouped_fwd(x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), *args, _stream())
    _lib.check(rc, "conv2d_grouped_fwd")
    return y


def conv2d_grouped_bwd_input(dy: Tensor, w: Tensor, x_shape, stride: int, pad: int, dil: int, groups: int) -> Tensor:
    _need(dy, bf16, "dy")
    _need(w, bf16, "w")
    dx = torch.empty(tuple(x_shape), dtype=bf16, device=dy.device)
    rc = _lib.load().cfhip_conv2d_grouped_bwd_input(dy.data_ptr(), w.data_ptr(), dx.data_ptr(),
                                                    *_gconv_args(x_shape, w, stride, pad, dil, groups), _stream())
    _lib.check(rc, "conv2d_grouped_bwd_input")
    return dx


def conv2d_grouped_bwd_weight(dy: Tensor, x: Tensor, dw: Tensor, accumulate: bool, bias_grad: Optional[Tensor],
                              bias_grad_accumulate: bool, stride: int, pad: int, dil: int, groups: int) -> None:
    """dw f32 [Cout, Cin / groups, kh, kw] (+)= the filter gradient, bias_grad f32 [Cout] (+)= dy summed over b, y, x"""
    _need(dy, bf16, "dy")
    _need(x, bf16, "x")
    _need(dw, f32, "dw")
    rc = _lib.load().cfhip_conv2d_grouped_bwd_weight(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), int(accumulate), _p(bias_grad),
                                                     int(bias_grad_accumulate), *_gconv_args(x.shape, dw, stride, pad, dil, groups),
                                                     _stream())
    _lib.check(rc, "conv2d_grouped_bwd_weight")


def conv3x3_wgrad_nhwc(dy_rows: Tensor, x_rows: Tensor, b: int, h: int, w: int, split_k: int = 1,
                       out: Optional[Tensor] = None, accumulate: bool = False, bias_grad: Optional[Tensor] = None,
                       bias_grad_accumulate: bool = False) -> Tensor:
    """dy_rows bf16 [B*H*W, Cout], x_rows bf16 [B*H*W, Cin] (NHWC) -> f32 [Cout, Cin, 3, 3] (the reference's filter
    layout), written or added to `out`."""
    _need(dy_rows, bf16, "dy_rows")
    _need(x_rows, bf16, "x_rows")
    m, cout = dy_rows.shape
    cin = x_rows.shape[1]
    if m != b * h * w or x_rows.shape[0] != m or not dy_rows.is_contiguous() or not x_rows.is_contiguous():
        raise ValueError("cfhip conv3x3_wgrad_nhwc: dense NHWC rows [B*H*W, C] expected for both operands")
    if bias_grad is not None:
        _need(bias_grad, f32, "bias_grad")
    if out is None:
        out, accumulate = torch.empty((cout, cin, 3, 3), dtype=f32, device=dy_rows.device), False
    else:
        _need(out, f32, "out")
        if out.numel() != cout * cin * 9 or not out.is_contiguous():
            raise ValueError("cfhip conv3x3_wgrad_nhwc: `out` must be a contiguous f32 [Cout, Cin, 3, 3]")
    if FLOP_COUNTER is not None:
        FLOP_COUNTER.flops["conv3x3"] += 2.0 * m * cout * 9 * cin
    lib = _lib.load()
    nbytes = lib.cfhip_conv3x3_wgrad_workspace(cin, cout, split_k)
    ws = torch.empty((nbytes // 4,), dtype=f32, device=dy_rows.device)
    _lib.check(lib.cfhip_conv3x3_wgrad_nhwc_bf16(dy_rows.data_ptr(), x_rows.data_ptr(), out.data_ptr(), int(accumulate),
                                                 _p(bias_grad), int(bias_grad_accumulate), b, h, w, cin, cout, split_k,
                                                 ws.data_ptr(), nbytes, _stream()), "conv3x3_wgrad_nhwc")
    return out


def conv3x3_pack_filters(w16: Tensor, rotate: bool, out: Optional[Tensor] = None) -> Tensor:
    """w16 bf16 [Cout, Cin, 3, 3] -> [Cout, 9*Cin] (k = (ky, kx, c)) or, rotate=True, [Cin, 9*Cout] with the taps rotated
    by 180 degrees (k = (ky, kx, co)): the filter matrices of conv3x3_nhwc for the forward / the input gradient."""
    _need(w16, bf16, "w16")
    if w16.dim() != 4 or tuple(w16.shape[2:]) != (3, 3) or not w16.is_contiguous():
        raise ValueError("cfhip conv3x3_pack_filters: contiguous bf16 [Cout, Cin, 3, 3] expected")
    cout, cin = w16.shape[0], w16.shape[1]
    if out is None:
        out = torch.empty((cin, 9 * cout) if rotate else (cout, 9 * cin), dtype=bf16, device=w16.device)
    _lib.check(_lib.load().cfhip_conv3x3_pack_filters(w16.data_ptr(), out.data_ptr(), cout, cin, int(rotate), _stream()),
               "conv3x3_pack_filters")
    return out


PACK_GROUP_MAX = 64  # problems per launch of conv3x3_pack_grouped (csrc/conv.hip: PK_MAX)


def conv3x3_pack_grouped(items) -> None:
    """items: (w16 bf16 [Cout, Cin, 3, 3], out, rotate) — `conv3x3_pack_filters` for all of them in ceil(len / 64) launches"""
    lib = _lib.load()
    for i in range(0, len(items), PACK_GROUP_MAX):
        flat = []
        for w16, out, rotate in items[i:i + PACK_GROUP_MAX]:
            if w16.dtype != bf16 or w16.dim() != 4 or tuple(w16.shape[2:]) != (3, 3) or not w16.is_contiguous() or out.dtype != bf16 \
                    or out.numel() != w16.numel():
                raise ValueError("cfhip conv3x3_pack_grouped: contiguous bf16 [Cout, Cin, 3, 3] filters and bf16 outputs of the same size expected")
            flat += [w16.data_ptr(), out.data_ptr(), w16.shape[0], w16.shape[1], int(bool(rotate))]
        table = (ctypes.c_int64 * len(flat))(*flat)
        _lib.check(lib.cfhip_conv3x3_pack_filters_grouped(ctypes.addressof(table), len(flat) // 5, _stream()), "conv3x3_pack_filters_grouped")


def sgemm_f32(a: Tensor, b: Tensor, *, a_trans: bool = False, b_trans: bool = False,
              alpha_dev: Optional[Tensor] = None, alpha: float = 1.0) -> Tensor:
    """fp32 C[m,n] = alpha * alpha_dev[0] * sum_k A(m,k) B(n,k) on dense f32 matrices (a: [M,K] or, a_trans, [K,M];
    b: [N,K] or, b_trans, [K,N]) -- the CLIP similarity logits and the products of their backward."""
    _need(a, f32, "a")
    _need(b, f32, "b")
    if a.dim() != 2 or b.dim() != 2 or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError("cfhip sgemm_f32: dense 2-D f32 operands expected")
    m, k = (a.shape[1], a.shape[0]) if a_trans else (a.shape[0], a.shape[1])
    n, kb = (b.shape[1], b.shape[0]) if b_trans else (b.shape[0], b.shape[1])
    if k != kb:
        raise ValueError(f"cfhip sgemm_f32: reduction dims differ ({k} vs {kb})")
    if alpha_dev is not None:
        _need(alpha_dev, f32, "alpha_dev")
    out = torch.empty((m, n), dtype=f32, device=a.device)
    _lib.check(_lib.load().cfhip_sgemm_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), m, n, k, a.stride(0), b.stride(0),
                                           int(a_trans), int(b_trans), _p(alpha_dev), float(alpha), _stream()), "sgemm_f32")
    return out


def dot_f32(a: Tensor, b: Tensor) -> Tensor:
    """f32 [1] = sum(a * b) over two dense f32 tensors of equal size."""
    _need(a, f32, "a")
    _need(b, f32, "b")
    if a.numel() != b.numel() or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError("cfhip dot_f32: dense f32 tensors of equal size expected")
    out = torch.zeros((1,), dtype=f32, device=a.device)
    _lib.check(_lib.load().cfhip_dot_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "dot_f32")
    return out


def spin(microseconds: int) -> None:
    """One idle wavefront on the current stream (stream self-check, see functional.distinct_stream)."""
    _lib.check(_lib.load().cfhip_spin(int(microseconds), _stream()), "spin")


def ema_update(ema: Tensor, p: Tensor, decay: float) -> None:
    """ema <- (1 - decay) * p + decay * ema over flat contiguous f32 buffers (bit-exact with the torch expression)."""
    _need(ema, f32, "ema")
    _need(p, f32, "p")
    if not ema.is_contiguous() or not p.is_contiguous() or ema.numel() != p.numel():
        raise ValueError("cfhip ema_update: contiguous f32 buffers of equal length expected")
    _lib.check(_lib.load().cfhip_ema_update(ema.data_ptr(), p.data_ptr(), p.numel(), float(1.0 - decay), float(decay),
                                            _stream()), "ema_update")


def geglu_fwd(vg: Tensor) -> Tensor:
    """vg bf16 [..., 2L] -> value * gelu(gate), bf16 [..., L]"""
    _need(vg, bf16, "vg")
    if not vg.is_contiguous() or vg.shape[-1] % 8:
        raise ValueError("cfhip geglu_fwd: contiguous [..., 2L] with L a multiple of 4 expected")
    l = vg.shape[-1] // 2
    out = torch.empty((*vg.shape[:-1], l), dtype=bf16, device=vg.device)
    _lib.check(_lib.load().cfhip_geglu_fwd(vg.data_ptr(), out.data_ptr(), vg.numel() // (2 * l), l, _stream()), "geglu_fwd")
    return out


def geglu_bwd(dy: Tensor, vg: Tensor) -> Tensor:
    _need(dy, bf16, "dy")
    _need(vg, bf16, "vg")
    l = vg.shape[-1] // 2
    dvg = torch.empty_like(vg)
    _lib.check(_lib.load().cfhip_geglu_bwd(dy.contiguous().data_ptr(), vg.data_ptr(), dvg.data_ptr(),
                                           vg.numel() // (2 * l), l, _stream()), "geglu_bwd")
    return dvg


def copy_strided(src: Tensor, dst: Tensor, batch: int, n: int, src_bs: int, dst_bs: int, src_off: int = 0,
                 dst_off: int = 0) -> None:
    """dst.flat[dst_off + b*dst_bs + i] = src.flat[src_off + b*src_bs + i] for b < batch, i < n (bf16)"""
    _need(src, bf16, "src")
    _need(dst, bf16, "dst")
    rc = _lib.load().cfhip_copy_strided_bf16(src.data_ptr() + 2 * src_off, dst.data_ptr() + 2 * dst_off, batch, n, src_bs,
                                             dst_bs, _stream())
    _lib.check(rc, "copy_strided_bf16")


def copy_strided2(src_a: Tensor, dst_a: Tensor, n_a: int, sa_bs: int, da_bs: int, src_b: Tensor, dst_b: Tensor, n_b: int, sb_bs: int,
                  db_bs: int, batch: int, *, src_a_off: int = 0, dst_a_off: int = 0, src_b_off: int = 0, dst_b_off: int = 0) -> None:
    """two `copy_strided` with the same batch count in one launch (a channel concatenation, or its backward split)"""
    for t, nm in ((src_a, "src_a"), (dst_a, "dst_a"), (src_b, "src_b"), (dst_b, "dst_b")):
        _need(t, bf16, nm)
    rc = _lib.load().cfhip_copy_strided2_bf16(src_a.data_ptr() + 2 * src_a_off, dst_a.data_ptr() + 2 * dst_a_off, n_a, sa_bs, da_bs,
                                              src_b.data_ptr() + 2 * src_b_off, dst_b.data_ptr() + 2 * dst_b_off, n_b, sb_bs, db_bs,
                                              batch, _stream())
    _lib.check(rc, "copy_strided2_bf16")


def q_sample(x: Tensor, noise: Tensor, t: Tensor, sqrt_ac: Tensor, sqrt_1mac: Tensor, out_dtype: torch.dtype = f32) -> Tensor:
    """x_t = sqrt_ac[t] * x + sqrt_1mac[t] * noise per sample (reference samplers/schema.py:94-108)"""
    for tt, nm in ((x, "x"), (noise, "noise"), (sqrt_ac, "sqrt_ac"), (sqrt_1mac, "sqrt_1mac")):
        _need(tt, f32, nm)
    if t.dtype != torch.int64 or not t.is_cuda:
        raise TypeError("cfhip q_sample: timesteps must be int64 on the device")
    x, noise = x.contiguous(), noise.contiguous()
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    b = x.shape[0]
    rc = _lib.load().cfhip_q_sample(x.data_ptr(), noise.data_ptr(), t.contiguous().data_ptr(), sqrt_ac.data_ptr(),
                                    sqrt_1mac.data_ptr(), out.data_ptr(), int(out_dtype == f32), b, x.numel() // b,
                                    _stream())
    _lib.check(rc, "q_sample")
    return out


def mse_loss(pred: Tensor, target: Tensor, grad_scale: float, want_grad: bool = True):
    """Returns (loss_sum f32 [1] = sum_b mean((pred_b - target_b)^2), dpred bf16 = grad_scale * d loss_sum / d pred)."""
    _need(pred, bf16, "pred")
    _need(target, f32, "target")
    pred, target = pred.contiguous(), target.contiguous()
    b = pred.shape[0]
    loss = torch.zeros((1,), dtype=f32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    rc = _lib.load().cfhip_mse_loss(pred.data_ptr(), target.data_ptr(), loss.data_ptr(), _p(dpred), b, pred.numel() // b,
                                    float(grad_scale), _stream())
    _lib.check(rc, "mse_loss")
    return loss, dpred


def diffusion_loss(pred: Tensor, target: Tensor, weight: Tensor, loss_type: str = "l2", want_grad: bool = True):
    """DDPMStep.loss_fn's per-sample objective: returns (per_sample f32 [B] = mean_inner f(pred - target), dpred bf16 =
    weight[b] * f' / inner); f = square ("l2") or abs ("l1")."""
    _need(pred, bf16, "pred")
    _need(target, f32, "target")
    _need(weight, f32, "weight")
    if loss_type not in ("l1", "l2"):
        raise ValueError(f"unrecognized loss '{loss_type}' occurred")
    pred, target, weight = pred.contiguous(), target.contiguous(), weight.contiguous()
    b = pred.shape[0]
    if weight.numel() != b or target.numel() != pred.numel():
        raise ValueError("cfhip diffusion_loss: shapes differ")
    per_sample = torch.zeros((b,), dtype=f32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    rc = _lib.load().cfhip_diffusion_loss(pred.data_ptr(), target.data_ptr(), weight.data_ptr(), per_sample.data_ptr(),
                                          _p(dpred), b, pred.numel() // b, 1 if loss_type == "l1" else 0, _stream())
    _lib.check(rc, "diffusion_loss")
    return per_sample, dpred


# ---------------------------------------------------------------------------------------------
# A12: dropout / DropPath (csrc/random.hip).  The generator state is (seed, offset): every call that draws random bits
# advances `offset` by the number of Philox counters it consumed, so that no two calls share bits and a backward pass
# can regenerate its mask from the (seed, offset) pair the forward recorded.
# ---------------------------------------------------------------------------------------------


class PhiloxState:
    """Process-wide counter-based generator state.  The stream starts from `torch.initial_seed()` + the rank of the process
    (so `torch.manual_seed` / the reference's `seed_everything` before the first draw seed it, and data-parallel ranks
    draw different masks); `manual_seed` (re)starts it explicitly.  (seed, offset) travel to the kernels as launch
    ARGUMENTS: a captured hipGraph would replay one frozen pair — the same masks every step — so drawing while a
    stream is capturing is an error (run stochastic modules eagerly: `TrainStep(use_graph=False)`)."""

    seed: Optional[int] = None  # None: derived from torch.initial_seed() + RANK at the first draw
    offset: int = 0

    @classmethod
    def manual_seed(cls, seed: int) -> None:
        cls.seed, cls.offset = int(seed) & 0xFFFFFFFFFFFFFFFF, 0

    @classmethod
    def take(cls, counters: int) -> Tuple[int, int]:
        """Reserve `counters` Philox counters; returns the (seed, offset) to hand to the kernel."""
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("cfhip: a dropout / DropPath mask was drawn during hipGraph capture: the (seed, offset) pair "
                               "would be frozen into the graph and every replay would apply the same mask; run models with "
                               "dropout or drop_path eagerly (TrainStep(use_graph=False))")
        if cls.seed is None:
            import os

            cls.seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (int(os.environ.get("RANK", "0")) + 1)) & 0xFFFFFFFFFFFFFFFF
        off = cls.offset
        cls.offset += int(counters)
        return cls.seed, off


def dropout(x: Tensor, p: float, *, seed: int = 0, offset: int = 0, mask: Optional[Tensor] = None,
            want_mask: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
    """y = keep ? x / (1 - p) : 0 (bf16 or f32, any shape).  `mask` (uint8, same number of elements) injects the keep
    mask instead of the Philox stream keyed by (seed, offset).  Returns (y, mask_out or None)."""
    if x.dtype not in (bf16, f32):
        raise TypeError(f"cfhip dropout: x must be bf16 or f32, got {x.dtype}")
    _need(x, x.dtype, "x")
    x = x.contiguous()
    y = torch.empty_like(x)
    if mask is not None:
        _need(mask, torch.uint8, "mask")
        if mask.numel() != x.numel():
            raise ValueError("cfhip dropout: mask must have one byte per element of x")
        mask = mask.contiguous()
    mask_out = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_mask else None
    rc = _lib.load().cfhip_dropout(x.data_ptr(), y.data_ptr(), int(x.dtype == f32), x.numel(), float(p), int(seed),
                                   int(offset), _p(mask), _p(mask_out), _stream())
    _lib.check(rc, "dropout")
    return y, mask_out


def drop_path_mask(batch: int, keep_prob: float, device: torch.device, *, seed: int = 0, offset: int = 0) -> Tensor:
    """mask[b] = floor(keep_prob + u_b) in {0, 1} (f32 [B]) — reference customs.py:439-441"""
    out = torch.empty((batch,), dtype=f32, device=device)
    rc = _lib.load().cfhip_drop_path_mask(out.data_ptr(), batch, float(keep_prob), int(seed), int(offset), _stream())
    _lib.check(rc, "drop_path_mask")
    return out


def drop_path(x: Tensor, mask: Tensor, keep_prob: float) -> Tensor:
    """y[b] = (x[b] / keep_prob) * mask[b] for a [B, ...] tensor (bf16 or f32); also its own backward on dy."""
    if x.dtype not in (bf16, f32):
        raise TypeError(f"cfhip drop_path: x must be bf16 or f32, got {x.dtype}")
    _need(x, x.dtype, "x")
    _need(mask, f32, "mask")
    x = x.contiguous()
    b = x.shape[0]
    if mask.numel() != b:
        raise ValueError("cfhip drop_path: one mask value per sample")
    y = torch.empty_like(x)
    rc = _lib.load().cfhip_drop_path(x.data_ptr(), y.data_ptr(), int(x.dtype == f32), mask.contiguous().data_ptr(),
                                     float(keep_prob), b, x.numel() // b, _stream())
    _lib.check(rc, "drop_path")
    return y


# ---------------------------------------------------------------------------------------------
# A16: tabular encoder gather (csrc/tabular.hip)
# ---------------------------------------------------------------------------------------------


def ml_encode_fwd(x: Tensor, plan: Tensor, tables: Optional[Tensor], out_dim: int) -> Tensor:
    """x f32 [B, F] -> merged_all f32 [B, out_dim]; `plan` int32 [out_dim, 6] on the device (see include/cfhip.h),
    `tables` int64 device array of embedding-table base pointers (or None when there is no embedding column)."""
    _need(x, f32, "x")
    _need(plan, torch.int32, "plan")
    b, f, xs = _mat(x, "x")
    out = torch.empty((b, out_dim), dtype=f32, device=x.device)
    rc = _lib.load().cfhip_ml_encode_fwd(x.data_ptr(), b, f, xs, plan.data_ptr(), out_dim, _p(tables), out.data_ptr(),
                                         _stream())
    _lib.check(rc, "ml_encode_fwd")
    return out


def ml_encode_indices(x: Tensor, cols: Tensor, dims: Tensor) -> Tensor:
    """int64 [B, K]: the reference's `EncodingResult.indices` (oob -> 0, truncation)"""
    _need(x, f32, "x")
    _need(cols, torch.int32, "cols")
    _need(dims, torch.int32, "dims")
    b, _, xs = _mat(x, "x")
    k = cols.numel()
    out = torch.empty((b, k), dtype=torch.int64, device=x.device)
    rc = _lib.load().cfhip_ml_encode_indices(x.data_ptr(), b, xs, cols.data_ptr(), dims.data_ptr(), k, out.data_ptr(),
                                             _stream())
    _lib.check(rc, "ml_encode_indices")
    return out


def ml_encode_bwd(dout: Tensor, x: Tensor, plan: Tensor, dtables: Optional[Tensor], want_dx: bool) -> Optional[Tensor]:
    _need(dout, f32, "dout")
    _need(x, f32, "x")
    b, f, xs = _mat(x, "x")
    dout = dout.contiguous()
    dx = torch.zeros((b, f), dtype=f32, device=x.device) if want_dx else None
    rc = _lib.load().cfhip_ml_encode_bwd(dout.data_ptr(), x.data_ptr(), b, f, xs, plan.data_ptr(), dout.shape[1],
                                         _p(dtables), _p(dx), _stream())
    _lib.check(rc, "ml_encode_bwd")
    return dx
