"""torch.autograd.Function wrappers: the operator seam of the drop-in.

Each Function replaces one ATen call of the reference's hot path with calls into libcfhip.so and
provides the matching hand-written backward.  Activations are bf16 (what the reference computes
under `mixed_precision="bf16"` autocast), parameters stay fp32 masters with a cached bf16 shadow,
parameter gradients are produced in fp32.

Parameter gradients are written by the backward kernels STRAIGHT into `param.grad` (allocated on
first use, accumulated afterwards) instead of being returned to autograd: that removes one full
read-modify-write pass per parameter and lets the gradient arena / DDP bucket views
(`optim.ParamArena`, `ddp.BucketedAllReduce`) be the kernels' destination.  Because autograd's own
post-accumulate hooks do not fire for such parameters, `grad_ready_callbacks` is invoked instead.
"""
import contextlib
import os
import threading
import weakref
from typing import Any, Callable, List, Optional, Sequence, Tuple

import math

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib as _lib_mod
from . import ops

bf16 = torch.bfloat16
f32 = torch.float32

# called as cb(param) right after a parameter's gradient has been written by a HIP backward
grad_ready_callbacks: List[Callable[[Tensor], None]] = []
# Called at the top of a backward that re-enters autograd (gradient checkpointing: `torch.autograd.grad` inside
# `_CheckpointFn.backward`), i.e. still in the OUTER graph task: whoever wants an end-of-backward callback of the whole
# pass (ddp.BucketedAllReduce) must queue it from here — queued from a gradient notification inside the nested task it
# would fire when that block's inner backward ends.
backward_entered_callbacks: List[Callable[[], None]] = []
# Weight gradients may be queued for a later grouped launch (fused.queue_linear_dw): whoever is about to consume ALL
# gradients of the pass from an end-of-backward callback of its own (ddp.BucketedAllReduce.finish) calls these first —
# the order in which autograd runs end-of-backward callbacks is the order they were queued in, which is not ours to pick.
deferred_grad_flushes: List[Callable[[], None]] = []


def notify_grad_ready(prm: Tensor) -> None:
    """A HIP backward kernel has been LAUNCHED that leaves `prm.grad` final for this pass: tell the listeners (DDP buckets,
    the optimizer's in-backward ranges).  Recorded by a launch plan (fused.StackPlan) together with the stream it fired on."""
    prm._cfhip_fresh = False
    rec = _lib_mod.RECORDER
    if rec is None:
        for cb in grad_ready_callbacks:
            cb(prm)
        return
    # recording: what the listeners launch themselves (an optimizer range update, a bucket's all-reduce) is THEIR business at
    # replay time too — the plan records the notification, not those launches
    _lib_mod.RECORDER = None
    try:
        for cb in grad_ready_callbacks:
            cb(prm)
    finally:
        _lib_mod.RECORDER = rec
    rec.append((2, prm, cur_stream()))


def rec_wait_stream(waiter: "torch.cuda.Stream", waited: "torch.cuda.Stream") -> None:
    waiter.wait_stream(waited)
    if _lib_mod.RECORDER is not None:
        _lib_mod.RECORDER.append((1, waiter.wait_stream, (waited,)))


def rec_wait_event(stream: "torch.cuda.Stream", ev: Any) -> None:
    stream.wait_event(ev)
    if _lib_mod.RECORDER is not None:
        _lib_mod.RECORDER.append((1, stream.wait_event, (ev,)))


def rec_record_event(ev: Any, stream: "torch.cuda.Stream") -> None:
    ev.record(stream)
    if _lib_mod.RECORDER is not None:
        _lib_mod.RECORDER.append((1, ev.record, (stream,)))


# ---------------------------------------------------------------------------------------------
# side HIP stream for the parameter-gradient kernels
# ---------------------------------------------------------------------------------------------
# In backward, dX = dY W (needed by the next layer: the critical path) and dW = dY^T X / db (needed
# only by the optimizer) are independent.  Issuing the dW work on a second HIP stream lets the two
# kernel queues interleave on the CUs: one kernel's store tail and partial last wave of workgroups
# are filled by the other's MFMA work (measured in profiles/).  Inside a hipGraph capture the side
# stream simply becomes a parallel branch of the graph.


# ROCclr multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams that share a queue
# run their kernels back to back.  Measured (profiles/r01/rccl_1rank_queues.log + the rocprofv3 queue column): once
# torch.distributed had created its own streams, the dW side stream landed on the compute stream's queue and the step
# lost its overlap (22.9 -> 26.8 ms).  So every helper stream is CHECKED: an idle wavefront on the new stream and on
# each stream it must overlap with, launched together, has to take the time of one; a stream that fails is parked and
# another one is created.  With the check the 1-rank RCCL step is 23.6 ms against 23.1 ms without DDP.  Raising
# GPU_MAX_HW_QUEUES to 8 makes the DDP step SLOWER (27.5 ms, three alternating runs), so the default is left alone.
_SPIN_US = 300
_rejected_streams: List["torch.cuda.Stream"] = []  # kept alive so that their queue slot stays taken
_stream_checks: List[dict] = []  # one record per distinct_stream() call: what stream_report() sums up


def _overlap(streams: List["torch.cuda.Stream"]) -> bool:
    import time

    for st in streams:  # the first launch on a stream creates its queue: not part of the measurement
        with torch.cuda.stream(st):
            ops.spin(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in streams:
        with torch.cuda.stream(st):
            ops.spin(_SPIN_US)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) < 1.6e-6 * _SPIN_US


def distinct_stream(others: List["torch.cuda.Stream"], tries: int = 12, priority: int = 0) -> "torch.cuda.Stream":
    """A new stream whose kernels run concurrently with those of every stream in `others` (and of the current one)."""
    base = [torch.cuda.current_stream()] + [st for st in others if st is not None]
    first = None
    for n in range(tries):
        cand = torch.cuda.Stream(priority=priority)
        first = first or cand
        # pairwise against each stream: robust to `others` that already alias one another
        if all(_overlap([st, cand]) for st in base):
            _stream_checks.append(dict(distinct=True, tries=n + 1, against=len(base)))
            return cand
        _rejected_streams.append(cand)
    import warnings

    _stream_checks.append(dict(distinct=False, tries=tries, against=len(base)))
    warnings.warn("cfhip: no helper stream on its own hardware queue after %d tries (GPU_MAX_HW_QUEUES=%s): "
                  "side-stream work will serialise with the compute stream" % (tries, os.environ.get("GPU_MAX_HW_QUEUES")))
    return first


def stream_report() -> dict:
    """What the helper-stream checks of this process found: `distinct` is False as soon as ONE helper stream (batch slice,
    weight-gradient lane, comm stream) had to share a hardware queue with a stream it must overlap with — the step then
    still computes the same numbers, serialised (22.9 -> 26.8 ms when it happened in round 1).  The budget: ROCclr
    multiplexes HIP streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; the step uses the caller's stream, two
    side lanes (SideStream.lanes) and — data parallel — one comm stream: four."""
    return dict(distinct=all(c["distinct"] for c in _stream_checks), helper_streams=len(_stream_checks),
                tries=max([c["tries"] for c in _stream_checks] or [0]), rejected=len(_rejected_streams),
                max_hw_queues=os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"))


# torch.cuda.current_stream() / torch.cuda.stream(...) walk four Python layers per call (9 / ~20 us); the side-stream
# plumbing below runs ~600 times per UNet step, on the thread that issues every launch.  Same torch entry points, called
# directly (tools/host_profile.py: 14 ms of 65 ms host time per step).
_get_cur = getattr(torch._C, "_cuda_getCurrentStream", None)
_set_cur = getattr(torch._C, "_cuda_setStream", None)
_get_dev = getattr(torch._C, "_cuda_getDevice", None)
_FAST_STREAMS = _get_cur is not None and _set_cur is not None and _get_dev is not None
_cuda_ok: Optional[bool] = None


def cuda_ok() -> bool:
    global _cuda_ok
    if _cuda_ok is None:
        _cuda_ok = torch.cuda.is_available()
    return _cuda_ok


def cur_stream() -> "torch.cuda.Stream":
    """torch.cuda.current_stream() of the current device"""
    if _FAST_STREAMS:
        sid, idx, typ = _get_cur(_get_dev())
        return torch.cuda.Stream(stream_id=sid, device_index=idx, device_type=typ)
    return torch.cuda.current_stream()


class on_stream:
    """`with torch.cuda.stream(side)` without the Python layers (same allocator semantics: tensors created inside belong
    to `side`)."""

    __slots__ = ("side", "prev")

    def __init__(self, side: "torch.cuda.Stream") -> None:
        self.side = side
        self.prev = None

    def __enter__(self) -> None:
        if _FAST_STREAMS:
            self.prev = _get_cur(_get_dev())
            _set_cur(stream_id=self.side.stream_id, device_index=self.side.device_index, device_type=self.side.device_type)
        else:
            self.prev = torch.cuda.current_stream()
            torch.cuda.set_stream(self.side)

    def __exit__(self, *exc: Any) -> None:
        if _FAST_STREAMS:
            _set_cur(stream_id=self.prev[0], device_index=self.prev[1], device_type=self.prev[2])
        else:
            torch.cuda.set_stream(self.prev)


# HIP stream priority per side lane (A/B knob: CFHIP_LANE_PRIORITY="0:-1,1:0"; negative = higher priority)
LANE_PRIORITY: dict = {}  # lane -> HIP stream priority; measured flat (profiles/r05, tools/gpu/lane_priority_ab.sh): a variable for A/B scripts, no environment switch since round 6


class SideStream:
    enabled = True
    # side streams: lane 0 carries the weight-gradient launches (and the forward's second batch slice), lane 1 the backward's
    # second batch slice.  TWO, not more: with the caller's stream and the comm stream of a data-parallel run that is the four
    # hardware queues ROCclr has by default — a third lane (three batch slices measured slower anyway) left the comm stream
    # without a queue of its own in every ProcessGroup run of round 3.  fused.py raises it when
    # CFHIP_FWD_HALVES / CFHIP_BWD_HALVES ask for more slices.
    lanes = 2
    heavy = True  # False: the dW GEMMs stay on the caller's stream, only the small reductions go aside
    streams: List[Optional["torch.cuda.Stream"]] = [None, None, None, None]
    keep: List[Tensor] = []  # operands produced on the main stream, alive until the join

    _join_queued = False

    @classmethod
    def get(cls, lane: int = 0) -> "torch.cuda.Stream":
        """The side stream of a lane; created (and checked for its own hardware queue) on first use."""
        lane = lane % max(1, cls.lanes)
        if cls.streams[lane] is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("cfhip: side streams must exist before a hipGraph capture starts (run one eager step)")
            cls.streams[lane] = distinct_stream(cls.streams, priority=LANE_PRIORITY.get(lane, 0))
        return cls.streams[lane]

    @classmethod
    def ensure(cls) -> None:
        for lane in range(max(1, cls.lanes)):
            cls.get(lane)

    @classmethod
    def _end_of_backward(cls) -> None:
        cls._join_queued = False
        cls.join()

    @classmethod
    def run(cls, fn: Callable[[], None], keep: Tuple[Tensor, ...] = (), lane: int = 0,
            wait: Tuple["torch.cuda.Stream", ...] = ()) -> None:
        """`fn` on the side stream of `lane`, ordered after the current stream and after every stream in `wait`
        (operands written by other streams, e.g. the second batch slice of the backward pass)."""
        if not cls.enabled or not cuda_ok() or (lane == 0 and not cls.heavy):
            for st in wait:
                cur_stream().wait_stream(st)
            fn()
            return
        if not cls._join_queued:
            # join automatically when the running backward pass ends, so that whoever reads `.grad`
            # afterwards on the caller's stream is ordered after the side-stream kernels
            try:
                torch.autograd.Variable._execution_engine.queue_callback(cls._end_of_backward)
                cls._join_queued = True
            except RuntimeError:  # not inside a backward pass: stay on the current stream
                for st in wait:
                    cur_stream().wait_stream(st)
                fn()
                return
        side = cls.get(lane)
        rec_wait_stream(side, cur_stream())  # everything issued so far is visible
        for st in wait:
            if st is not side:
                rec_wait_stream(side, st)
        with on_stream(side):
            fn()
        cls.keep.extend(keep)

    @classmethod
    def fork(cls, lane: int = 0) -> Optional["torch.cuda.Stream"]:
        """A side stream that has waited for the current stream (for work the caller joins itself)."""
        if not cls.enabled or not cuda_ok():
            return None
        side = cls.get(lane)
        rec_wait_stream(side, cur_stream())
        return side

    @classmethod
    def queue_join(cls) -> None:
        """A replayed backward (fused.StackPlan) put work on the side lanes without going through `run`: make sure the
        running backward pass ends with the join."""
        if not cls._join_queued:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(cls._end_of_backward)
                cls._join_queued = True
            except RuntimeError:
                pass

    @classmethod
    def join(cls) -> None:
        """Make the current stream wait for the side stream (call before consuming parameter grads)."""
        for side in cls.streams:
            if side is not None:
                cur_stream().wait_stream(side)
        cls.keep.clear()


# ---------------------------------------------------------------------------------------------
# Taped composite nodes (round 5): a whole sub-module as ONE autograd node
# ---------------------------------------------------------------------------------------------
# The composed module paths (the UNet's residual block and spatial transformer: residual.py:154-253, mixed_stacks/api.py:766-893)
# are chains of the Functions of this file.  Through autograd each link costs a `Function.apply` on the way in and an engine
# dispatch on the way back (8-9 us of host time each, ~600 links per zoo-UNet step whose forward is bound by the host's issue
# rate) and every fan-in is an ATen add.  Inside a taped node the SAME static `forward` / `backward` of every Function run
# against a plain context object: `TapedFn.forward` records (Function, context, argument slots) while the module's own forward
# code runs, `TapedFn.backward` walks the record in reverse, sums fan-ins with `cfhip_add_bf16` and lets LayerNorm's backward
# kernel take the gradient its input has already received as its `dx_add` operand.  Nothing is re-implemented per module.
# What the tape cannot see is a plain torch op BETWEEN two Functions (its result would silently drop out of the gradient):
# views of taped tensors are detected and end the attempt (`TapeBreak`: the module runs its composed path from then on); the
# modules that use the tape only do so for configurations whose op sequence is known to consist of Functions
# (tests/test_gpu_unet.py compares every gradient of both node types with the composed path).
# MEASURED (profiles/r05/unet_taped_nodes.txt): the zoo UNet's graph shrinks from 1 201 to 751 autograd nodes, the host's backward
# from ~28 to ~24-29 ms, its forward GROWS from 14.5 to 17-18 ms (collecting a node's parameters and slots costs more than the
# `apply` calls it saves) — and neither matters: the GPU needs 19.1 ms for the forward and 39.8 ms for the backward either way
# (events around the phases), so the step is bound by the queues in BOTH phases, 59.7 vs 59.6-60.0 ms.  Off by default.

TAPED_NODES = [os.environ.get("CFHIP_TAPED_NODES", "0") == "1"]  # built, measured, not selected: see the note below
_TAPE = threading.local()  # `.tape`: the tape that records on THIS thread (a forward on another thread is none of its business)


class TapeBreak(RuntimeError):
    """the sub-module's forward did something the tape cannot differentiate"""


class _TapeCtx:
    """what a Function's static `forward` / `backward` get in place of autograd's context while a tape records"""

    def __init__(self, needs: Tuple[bool, ...]):
        self.needs_input_grad = needs
        self.saved_tensors: Tuple[Any, ...] = ()

    def save_for_backward(self, *tensors: Any) -> None:
        self.saved_tensors = tensors

    def set_materialize_grads(self, value: bool) -> None:
        pass

    def mark_non_differentiable(self, *tensors: Any) -> None:
        pass


class _Tape:
    def __init__(self, inputs: Tuple[Any, ...]):
        self.n_inputs = len(inputs)
        self.by_id = {id(t): i for i, t in enumerate(inputs) if isinstance(t, Tensor) and t.requires_grad}
        self.inputs = inputs  # keeps the ids stable
        self.entries: List[Tuple[Any, _TapeCtx, Tuple[Any, ...], int]] = []
        self.next_slot = len(inputs)
        self.out_slot = -1
        self.used = False
        self.token = object()  # what taped results carry (not the tape itself: a result saved by its own link would close a cycle)

    def slot(self, t: Tensor) -> Optional[Tuple[int, Optional[torch.Size]]]:
        """(slot, None) of a node input or a taped result; (slot, parameter shape) of a contiguous view of a whole parameter;
        None for a constant"""
        s = getattr(t, "_cfhip_slot", None)
        if s is not None and s[0] is self.token:
            return s[1], None
        i = self.by_id.get(id(t))
        if i is not None:
            return i, None
        if t._is_view():
            base = t._base
            i = None if base is None else self.by_id.get(id(base))
            if (i is not None and base.is_leaf and base.numel() == t.numel() and t.is_contiguous() and base.is_contiguous()
                    and t.data_ptr() == base.data_ptr()):
                return i, base.shape  # e.g. a [Cout, Cin, 1, 1] filter seen as the matrix of its GEMM (cf. whole_param)
            if base is not None and (i is not None or (getattr(base, "_cfhip_slot", None) or (None,))[0] is self.token):
                raise TapeBreak("a view of a taped tensor reached a Function: the tape has no gradient rule for the view")
        return None

    def record(self, fn: Any, args: Tuple[Any, ...]) -> Any:
        slots = tuple(self.slot(a) if isinstance(a, Tensor) else None for a in args)
        ctx = _TapeCtx(tuple(sl is not None for sl in slots))
        out = fn.forward(ctx, *args)
        if any(sl is not None for sl in slots):
            if not isinstance(out, Tensor):
                raise TapeBreak(f"{fn.__name__} returns {type(out).__name__}: only single-tensor Functions are taped")
            if any(out is a for a in args):  # an identity link: its result needs an identity of its own
                out = out.view_as(out)
            out._cfhip_slot = (self.token, self.next_slot)
            metas = tuple(None if sl is None else (sl[0], sl[1], a.dtype, a.shape) for sl, a in zip(slots, args))
            self.entries.append((fn, ctx, metas, self.next_slot))
            self.next_slot += 1
        return out

    def backward(self, dy: Tensor) -> List[Optional[Tensor]]:
        if self.used:
            raise RuntimeError("cfhip: a taped node was differentiated twice (retain_graph / double backward); set "
                               "CFHIP_TAPED_NODES=0 or functional.TAPED_NODES[0] = False for such a graph")
        self.used = True
        grads: dict = {self.out_slot: dy}
        entries = self.entries
        while entries:
            fn, ctx, metas, out_slot = entries.pop()  # popped: what the link saved is freed as the walk passes it
            g = grads.pop(out_slot, None)
            if g is None:
                continue
            first = metas[0]
            if getattr(fn, "folds_dx_add", False) and first is not None and first[1] is None:
                pending = grads.get(first[0])
                if pending is not None and pending.dtype == bf16 and pending.shape == first[3] and pending.is_contiguous():
                    ctx.dx_add = grads.pop(first[0])  # the kernel adds it: the link's dx then IS the sum so far
            gin = fn.backward(ctx, g)
            if not isinstance(gin, tuple):
                gin = (gin,)
            for meta, ga in zip(metas, gin):
                if meta is None or ga is None:
                    continue
                slot, param_shape, dtype, shape = meta
                if ga.shape != shape:
                    raise RuntimeError(f"cfhip taped node: {fn.__name__}.backward returned {tuple(ga.shape)} for an input of {tuple(shape)}")
                if ga.dtype != dtype:
                    ga = ga.to(dtype)  # what autograd's engine does between two nodes
                if param_shape is not None:
                    ga = ga.reshape(param_shape)
                prev = grads.get(slot)
                grads[slot] = ga if prev is None else _sum_grads(prev, ga)
        self.inputs = ()
        return [grads.get(i) for i in range(self.n_inputs)]


def _sum_grads(a: Tensor, b: Tensor) -> Tensor:
    if a.dtype == bf16 and b.dtype == bf16 and a.is_cuda and a.shape == b.shape:
        return AddFn.forward(None, a, b)  # a new tensor: either operand may be somebody else's gradient too
    return a + b


def _apply(fn: Any, *args: Any) -> Any:
    """`fn.apply(*args)`, or — inside a taped node — the Function's forward run directly and put on the tape"""
    tape = getattr(_TAPE, "tape", None)
    if tape is None:
        return fn.apply(*args)
    if not getattr(fn, "tapeable", True):
        raise TapeBreak(f"{fn.__name__} cannot run inside a taped node")
    return tape.record(fn, args)


class TapedFn(Function):
    @staticmethod
    def forward(ctx: Any, run: Callable[[], Tensor], *tensors: Any) -> Tensor:
        tape = _Tape(tensors)
        _TAPE.tape = tape
        try:
            out = run()
        finally:
            _TAPE.tape = None
        s = getattr(out, "_cfhip_slot", None)
        if s is None or s[0] is not tape.token:
            raise TapeBreak("the sub-module's result is not the result of a taped Function")
        tape.out_slot = s[1]
        ctx.tape = tape
        return out

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        return (None,) + tuple(ctx.tape.backward(dy))


def run_taped(run: Callable[[], Tensor], tensors: Tuple[Any, ...]) -> Tensor:
    """`run()` as one autograd node over `tensors` (the activations it reads and every parameter it uses); plainly when a tape is
    already recording (nested sub-modules join the outer node), without grad mode, or when nothing asks for a gradient."""
    if getattr(_TAPE, "tape", None) is not None or not torch.is_grad_enabled():
        return run()
    return TapedFn.apply(run, *tensors)


# ---------------------------------------------------------------------------------------------
# bf16 shadows of fp32 master parameters
# ---------------------------------------------------------------------------------------------


def shadow_bf16(param: Tensor) -> Tensor:
    """bf16 copy of an fp32 parameter, re-cast only when the parameter changed.

    `optim.ParamArena` installs arena-backed shadows that its fused Adam kernel refreshes in the
    same pass that updates the master weights; otherwise the copy is keyed on `param._version`.
    """
    if param.dtype == bf16:
        return param.detach()
    sh = getattr(param, "_cfhip_shadow", None)
    ver = param._version
    if sh is not None and getattr(param, "_cfhip_shadow_version", None) == ver and sh.device == param.device:
        return sh
    src = param.detach()
    if sh is None or sh.shape != src.shape or sh.device != src.device:
        sh = torch.empty(src.shape, dtype=bf16, device=src.device)
    ops.to_bf16(src.contiguous(), out=sh)
    try:
        param._cfhip_shadow = sh
        param._cfhip_shadow_version = ver
    except Exception:  # non-leaf views etc.: just do not cache
        pass
    return sh


def _is_direct(param: Optional[Tensor]) -> bool:
    return param is not None and param.is_leaf and param.requires_grad and param.dtype == f32


def whole_param(t: Optional[Tensor]) -> Optional[Tensor]:
    """`t` itself, or — when `t` is a contiguous view of a WHOLE leaf parameter (a [Cout, Cin, 1, 1] filter seen as the
    [Cout, Cin] matrix of the GEMM it is) — that parameter: its gradient can then be written straight into `.grad` (and
    run on the side stream) instead of travelling back through autograd's view chain."""
    if t is None or not t._is_view():
        return t
    # A view taken WITHOUT grad mode is a leaf that does not require grad.  Inside a taped node (TapedFn.forward runs without grad
    # mode) that is the normal case and the view stands for its parameter; anywhere else it is a weight somebody froze on purpose
    # (a view cached under torch.no_grad()): it must not resolve to its base and silently receive gradients / optimizer updates
    # (ADVICE r5).
    if t.is_leaf and getattr(_TAPE, "tape", None) is None:
        return t
    base = t._base
    if (base is not None and base.is_leaf and base.requires_grad and base.dtype == f32 and base.numel() == t.numel()
            and t.is_contiguous() and base.is_contiguous() and t.data_ptr() == base.data_ptr()):
        return base
    return t


def grad_buffer(param: Tensor, zero: bool = False) -> Tensor:
    """The tensor a backward kernel writes `param`'s gradient into when `param.grad is None`: the parameter's slot of
    its `optim.ParamArena` when it lives in one (the reference trainer's `optimizer.zero_grad()` sets `.grad` to None
    every step; the bucketed all-reduce and the fused Adam read the arena), a new tensor otherwise.  Contents are
    undefined unless `zero`."""
    arena = getattr(param, "_cfhip_arena", None)
    if arena is not None:
        buf = arena.grad_view(param)
        if zero:
            buf.zero_()
        return buf
    make = torch.zeros if zero else torch.empty
    return make(param.shape, dtype=f32, device=param.device)


def write_param_grad(param: Tensor, compute: Callable[[Tensor, bool], None]) -> None:
    """`compute(out, accumulate)` must write (accumulate=False) or add (True) the f32 gradient."""
    g = param.grad
    if g is None:
        buf = grad_buffer(param)
        compute(buf, False)
        param.grad = buf
        param._cfhip_fresh = False
    else:
        fresh = getattr(param, "_cfhip_fresh", False)
        compute(g, not fresh)
        if fresh:
            param._cfhip_fresh = False
    for cb in grad_ready_callbacks:
        cb(param)


def write_param_grad_pair(a: Tensor, b: Tensor, compute: Callable[[Tensor, Tensor, bool], None]) -> bool:
    """Two gradients that ONE kernel writes with one accumulate flag (LayerNorm's dgamma / dbeta): `compute(out_a, out_b,
    accumulate)`.  False (nothing done) when one of the two has to be accumulated and the other written."""
    def accumulates(prm: Tensor) -> bool:
        return prm.grad is not None and not getattr(prm, "_cfhip_fresh", False)

    acc = accumulates(a)
    if acc != accumulates(b):
        return False
    for prm in (a, b):
        if prm.grad is None:
            prm.grad = grad_buffer(prm)
    compute(a.grad, b.grad, acc)
    for prm in (a, b):
        prm._cfhip_fresh = False
        for cb in grad_ready_callbacks:
            cb(prm)
    return True


def _as_rows(x: Tensor) -> Tensor:
    """[..., D] -> 2-D view with a contiguous last dim, keeping f32 or bf16 as is."""
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype not in (bf16, f32):
        x2 = x2.float()
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    return x2


def _as_bf16_2d(x: Tensor) -> Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype != bf16:
        x2 = ops.to_bf16(x2.float() if x2.dtype != f32 else x2)
    if x2.stride(-1) != 1 or (x2.dim() == 2 and x2.stride(0) % 8 != 0 and x2.shape[0] > 1):
        x2 = x2.contiguous()
    return x2


# ---------------------------------------------------------------------------------------------
# Linear (K1): y = x W^T + b, optional fused GELU / residual-add epilogue
# ---------------------------------------------------------------------------------------------

ACT_NONE, ACT_GELU, ACT_QGELU = 0, 1, 2


def _all_direct(weight: Tensor, bias: Optional[Tensor]) -> bool:
    """Every gradient this op owes is written straight into an arena-owned `.grad` (nothing is returned to autograd)."""
    if weight.requires_grad and not _is_direct(weight):
        return False
    return bias is None or not bias.requires_grad or _is_direct(bias)


def _linear_param_grads(dy2: Tensor, x2: Tensor, weight: Tensor, bias: Optional[Tensor],
                        weight_direct: bool, bias_direct: bool) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """dW = dy^T x, db = colsum(dy): direct-to-.grad when possible, returned otherwise."""
    gw = gb = None
    n, k, m = dy2.shape[1], x2.shape[1], x2.shape[0]  # (the parameter may be a [n, k, 1, 1] filter: whole_param)
    fast_ok = dy2.shape[1] % 8 == 0 and x2.shape[1] % 8 == 0 and dy2.stride(0) % 8 == 0 and x2.stride(0) % 8 == 0
    # split-K lives on the MFMA path only; narrow tabular heads (3 classes, 10 features) take the shape-agnostic kernel
    split = ops.pick_split_k(n, k, m) if fast_ok else 1

    def dw_into(out: Tensor, acc: bool) -> None:
        ops.gemm(dy2, x2, a_trans=True, b_trans=True, out=out.view(n, k), accumulate=acc, split_k=split)

    if (weight.requires_grad and weight_direct and bias is not None and bias.requires_grad and bias_direct
            and fast_ok):
        from .fused import _dw_db  # one launch for dW and db

        _dw_db(weight, bias, dy2, x2)
        return None, None
    if weight.requires_grad:
        if weight_direct:
            write_param_grad(weight, dw_into)
        else:
            gw = torch.empty(weight.shape, dtype=f32, device=dy2.device)
            dw_into(gw, False)
    if bias is not None and bias.requires_grad:
        if bias_direct:
            write_param_grad(bias, lambda out, acc: ops.colsum(dy2, out=out.view(-1), accumulate=acc))
        else:
            gb = ops.colsum(dy2).view(bias.shape)
    return gw, gb


# ---- fan-in links (round 6; VERDICT r5 #3b "fan-in adds inside the consumer") -----------------------------------------------------------
# `net = f(LN(net)) + net` with the add riding in the last GEMM's epilogue (the UNet's SpatialTransformerBlock, three times per block):
# in backward the GEMM's Function returns dY for the residual operand, the LayerNorm's returns dx, and autograd's engine sums the two with an
# ATen add (45 launches and 1.7 GB of traffic per 64^2 x 8 step).  Inside a `fanin_links()` scope a LayerNormFn that sees a bf16 input
# requiring a gradient leaves a link behind; a LinearFn whose `residual` operand IS that tensor takes the link, and in backward hands its dY
# to the link instead of to autograd (None for the residual); the LayerNorm backward — which runs later, it sits upstream of that GEMM —
# passes it to its kernel as `dx_add`: one rounding instead of two, no add launch.  The scope is the module's promise that the GEMM consumes
# (a function of) that LayerNorm's output, i.e. that the LayerNorm backward runs after, and whenever, the GEMM's does.
FANIN_LINKS = True  # (A/B: tools set the attribute)


class FanInLink:
    __slots__ = ("pending", "__weakref__")

    def __init__(self) -> None:
        self.pending: Optional[Tensor] = None


class _FanInState(threading.local):
    def __init__(self) -> None:
        self.active = 0
        self.entries: List[Tuple[Any, FanInLink]] = []


_FANIN = _FanInState()


@contextlib.contextmanager
def fanin_links():
    if not FANIN_LINKS:
        yield
        return
    _FANIN.active += 1
    try:
        yield
    finally:
        _FANIN.active -= 1
        if _FANIN.active == 0:
            _FANIN.entries.clear()


def _fanin_offer(ctx: Any, x: Tensor) -> None:
    """LayerNormFn.forward: leave a link for the GEMM that will add `x` back"""
    ctx.link = None
    if _FANIN.active and x.dtype == bf16 and getattr(_TAPE, "tape", None) is None and ctx.needs_input_grad[0]:
        ctx.link = FanInLink()
        _FANIN.entries.append((weakref.ref(x), ctx.link))
        del _FANIN.entries[:-4]


def _fanin_take(residual: Optional[Tensor]) -> Optional[FanInLink]:
    """LinearFn.forward: the link of the LayerNorm that read this very tensor, if one is waiting"""
    if residual is None or not _FANIN.active or residual.dtype != bf16:
        return None
    for i, (ref, link) in enumerate(_FANIN.entries):
        if ref() is residual:
            del _FANIN.entries[i]
            return link
    return None


class LinearFn(Function):
    """Replaces F.linear (reference customs.py:89, attentions.py:214) incl. backward."""

    @staticmethod
    def forward(ctx: Any, x: Tensor, weight: Tensor, bias: Optional[Tensor], act: int,
                residual: Optional[Tensor], out_f32: bool) -> Tensor:
        x2 = _as_bf16_2d(x)
        wshape = weight.shape
        weight, bias = whole_param(weight), whole_param(bias)
        w16 = shadow_bf16(weight).view(wshape)
        bias_f = None if bias is None else bias.detach().reshape(-1).contiguous()
        m, n = x2.shape[0], wshape[0]
        pre = None
        if act in (ACT_GELU, ACT_QGELU):
            pre = torch.empty((m, n), dtype=bf16, device=x2.device)
            y = ops.gemm(x2, w16, bias=bias_f, epilogue=ops.EPI_GELU if act == ACT_GELU else ops.EPI_QGELU,
                         aux_out=pre)
        elif residual is not None:
            # the residual operand keeps its dtype: an f32 residual stream gives an f32 sum
            r2 = _as_rows(residual)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
            y = ops.gemm(x2, w16, bias=bias_f, epilogue=ops.EPI_RESIDUAL, aux_in=r2, out_dtype=r2.dtype)
        else:
            y = ops.gemm(x2, w16, bias=bias_f, out_dtype=f32 if out_f32 else bf16)
        ctx.save_for_backward(x2, w16, pre)
        ctx.weight, ctx.bias = weight, bias
        ctx.n = n
        ctx.has_residual = residual is not None
        ctx.link = _fanin_take(residual) if residual is not None and y.dtype == bf16 else None
        ctx.x_shape = x.shape
        ctx.in_dtype = x.dtype
        ctx.act = act
        return y.view(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        x2, w16, pre = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        dy2 = _as_bf16_2d(dy)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        d_res = dy2.view(*ctx.x_shape[:-1], ctx.n) if ctx.has_residual and ctx.needs_input_grad[4] else None
        # (autograd casts d_res to the residual's dtype; the gradient stream itself stays bf16)
        link = getattr(ctx, "link", None)
        if link is not None and d_res is not None:  # the LayerNorm that read the residual operand adds it inside its backward kernel
            link.pending, d_res = dy2, None
        if pre is not None:
            dy2 = ops.gelu_bwd(dy2, pre) if ctx.act == ACT_GELU else ops.quick_gelu_bwd(dy2, pre)
        gw = gb = None
        if _all_direct(weight, bias):
            # gradients that land straight in `.grad` do not go back through autograd: they run on the side stream,
            # beside the dX GEMM below (issued first so that the side stream does not wait for dX)
            from . import fused  # (round 3) weight gradients of stand-alone Linear layers join the grouped launches

            if not fused.queue_linear_dw(weight, bias, dy2, x2):
                SideStream.run(lambda: _linear_param_grads(dy2, x2, weight, bias, True, True), (dy2, x2))
            dx = ops.gemm(dy2, w16, b_trans=True).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        else:
            dx = None
            if ctx.needs_input_grad[0]:
                dx = ops.gemm(dy2, w16, b_trans=True).view(ctx.x_shape)  # bf16; autograd casts if needed
            gw, gb = _linear_param_grads(dy2, x2, weight, bias, _is_direct(weight), _is_direct(bias))
        return dx, gw, gb, None, d_res, None


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, *, act: int = ACT_NONE,
           residual: Optional[Tensor] = None, out_f32: bool = False) -> Tensor:
    return _apply(LinearFn, x, weight, bias, act, residual, out_f32)


def qkv_weights_adjacent(wq: Tensor, wk: Tensor, wv: Tensor) -> bool:
    """Three [D, C] parameters laid out back to back in ONE `optim.ParamArena` (registration order to_q, to_k, to_v, no bias
    in between: the reference's CrossAttention, attentions.py:505-512): their values, gradient slots and bf16 shadows are
    then each one contiguous [3 D, C] matrix."""
    ar = getattr(wq, "_cfhip_arena", None)
    if ar is None or getattr(wk, "_cfhip_arena", None) is not ar or getattr(wv, "_cfhip_arena", None) is not ar:
        return False
    if wq.dim() != 2 or wq.shape != wk.shape or wq.shape != wv.shape or not (wq.requires_grad and wk.requires_grad and wv.requires_grad):
        return False
    n = wq.numel()
    if wk.data_ptr() != wq.data_ptr() + 4 * n or wv.data_ptr() != wk.data_ptr() + 4 * n:
        return False
    sh = [getattr(w, "_cfhip_shadow", None) for w in (wq, wk, wv)]
    return all(t is not None for t in sh) and sh[1].data_ptr() == sh[0].data_ptr() + 2 * n and sh[2].data_ptr() == sh[1].data_ptr() + 2 * n


class QKVLinearFn(Function):
    """to_q(x) | to_k(x) | to_v(x) of a self-attention whose three bias-free projection weights are adjacent in the arena
    (`qkv_weights_adjacent`) as ONE GEMM against the [3 D, C] matrix they form: output [.., 3 D] packed for
    `packed_self_attention`.  Backward: one dX GEMM (instead of three + two adds of the partial input gradients) and one
    dW GEMM straight into the three adjacent gradient slots (instead of three GEMMs + three split-K reduces).  The UNet's 30
    self-attention modules: ~300 fewer launches per step."""

    @staticmethod
    def forward(ctx: Any, x: Tensor, wq: Tensor, wk: Tensor, wv: Tensor) -> Tensor:
        x2 = _as_bf16_2d(x)
        d, c = wq.shape
        w16 = shadow_bf16(wq)
        shadow_bf16(wk)
        shadow_bf16(wv)
        w3 = torch.as_strided(w16, (3 * d, c), (c, 1))
        y = ops.gemm(x2, w3)
        ctx.save_for_backward(x2, w3)
        ctx.prm, ctx.x_shape = (wq, wk, wv), x.shape
        return y.view(*x.shape[:-1], 3 * d)

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        x2, w3 = ctx.saved_tensors
        prm = ctx.prm
        d, c = prm[0].shape
        dy2 = _as_bf16_2d(dy)

        def param_grads() -> None:
            for w in prm:
                if w.grad is None:
                    w.grad = grad_buffer(w)
                    w._cfhip_fresh = True
            acc = [not getattr(w, "_cfhip_fresh", False) for w in prm]
            n = d * c
            together = (acc[0] == acc[1] == acc[2] and prm[1].grad.data_ptr() == prm[0].grad.data_ptr() + 4 * n
                        and prm[2].grad.data_ptr() == prm[1].grad.data_ptr() + 4 * n and prm[0].grad.is_contiguous())
            if together:
                g3 = torch.as_strided(prm[0].grad, (3 * d, c), (c, 1))
                ops.gemm(dy2, x2, a_trans=True, b_trans=True, out=g3, accumulate=acc[0], split_k=ops.pick_split_k(3 * d, c, x2.shape[0]))
            else:  # a gradient slot that is not the arena's (somebody replaced `.grad`), or mixed write / accumulate states
                for i, w in enumerate(prm):
                    ops.gemm(dy2[:, i * d:(i + 1) * d], x2, a_trans=True, b_trans=True, out=w.grad.view(d, c), accumulate=acc[i],
                             split_k=ops.pick_split_k(d, c, x2.shape[0]))
            for w in prm:
                w._cfhip_fresh = False
                for cb in grad_ready_callbacks:
                    cb(w)

        SideStream.run(param_grads, (dy2, x2))
        dx = ops.gemm(dy2, w3, b_trans=True).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        return dx, None, None, None


def qkv_linear(x: Tensor, wq: Tensor, wk: Tensor, wv: Tensor) -> Tensor:
    return _apply(QKVLinearFn, x, wq, wk, wv)


# ---------------------------------------------------------------------------------------------
# LayerNorm (K5)
# ---------------------------------------------------------------------------------------------


class LayerNormFn(Function):
    """Replaces nn.LayerNorm.forward (reference norms.py:88-89 via NormFactory)."""

    folds_dx_add = True  # a taped node hands the backward kernel the gradient x already has (`ctx.dx_add`, bf16, x's shape)

    @staticmethod
    def forward(ctx: Any, x: Tensor, weight: Tensor, bias: Tensor, eps: float, out_f32: bool = False) -> Tensor:
        x2 = _as_rows(x)
        gamma = weight.detach().contiguous()
        beta = bias.detach().contiguous()
        y, mean, rstd = ops.layernorm_fwd(x2, gamma, beta, eps, out_f32=bool(out_f32) and x2.dtype == f32)
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.weight, ctx.bias = weight, bias
        ctx.x_shape, ctx.in_dtype = x.shape, x.dtype
        _fanin_offer(ctx, x)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        x2, gamma, mean, rstd = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        dy2 = _as_bf16_2d(dy)
        wd, bd = _is_direct(weight), _is_direct(bias)
        gw = gb = None
        res: list = []
        dx_add = getattr(ctx, "dx_add", None)  # inside a taped node: the gradient x has already received from its other readers
        link = getattr(ctx, "link", None)
        if link is not None and link.pending is not None:  # fan-in link: the dY of the GEMM that added x back (see fanin_links)
            if dx_add is None and link.pending.shape == x2.shape:
                dx_add, link.pending = link.pending, None
            else:
                raise RuntimeError("cfhip: a fan-in link delivered a gradient the LayerNorm backward cannot take "
                                   f"({tuple(link.pending.shape)} for rows {tuple(x2.shape)})")
        if dx_add is not None:
            dx_add = dx_add.view(-1, dx_add.shape[-1])
        if wd and bd and write_param_grad_pair(
                weight, bias, lambda ga, gb_, acc: res.append(ops.layernorm_bwd(
                    dy2, x2, gamma, mean, rstd, dx_add=dx_add, dgamma=ga.view(-1), dbeta=gb_.view(-1), accumulate=acc)[0])):
            dx = res[0]  # the kernel reduced dgamma / dbeta straight into `.grad` (round 1-2: two device copies per call)
        else:
            dx, dg, db = ops.layernorm_bwd(dy2, x2, gamma, mean, rstd, dx_add=dx_add)
            if wd:
                write_param_grad(weight, lambda out, acc: out.add_(dg.view(out.shape)) if acc else out.copy_(dg.view(out.shape)))
            elif weight.requires_grad:
                gw = dg.view(weight.shape)
            if bd:
                write_param_grad(bias, lambda out, acc: out.add_(db.view(out.shape)) if acc else out.copy_(db.view(out.shape)))
            elif bias.requires_grad:
                gb = db.view(bias.shape)
        dx = dx.view(ctx.x_shape) if ctx.needs_input_grad[0] else None  # bf16; autograd casts if needed
        return dx, gw, gb, None, None


class LayerNorm4dFn(Function):
    """The reference's custom `LN.forward` on 4-D inputs (norms.py:30-46): per-sample mean / unbiased std over C*H*W, eps added to the
    standard deviation, per-channel affine — cfhip_layernorm4d_fwd / _bwd (csrc/norm.hip)."""

    @staticmethod
    def forward(ctx: Any, x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], eps: float) -> Tensor:
        x2 = to_nchw(x if x.dtype == bf16 else ops.to_bf16(x.float().contiguous())).contiguous()
        w = None if weight is None else weight.detach().float().contiguous().view(-1)
        b = None if bias is None else bias.detach().float().contiguous().view(-1)
        y, mean, std = ops.layernorm4d_fwd(x2, w, b, eps)
        ctx.save_for_backward(x2, mean, std)
        ctx.w, ctx.eps, ctx.affine = w, float(eps), weight is not None
        ctx.shapes = (None if weight is None else weight.shape, None if bias is None else bias.shape)
        return y

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        x2, mean, std = ctx.saved_tensors
        dy2 = to_nchw(dy if dy.dtype == bf16 else ops.to_bf16(dy.float().contiguous())).contiguous()
        dw = db = None
        want_pg = ctx.affine and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        if want_pg:
            dw = torch.empty((x2.shape[1],), dtype=f32, device=x2.device)
            db = torch.empty_like(dw)
        dx = ops.layernorm4d_bwd(dy2, x2, ctx.w, mean, std, ctx.eps, want_dx=ctx.needs_input_grad[0], dweight=dw, dbias=db)
        if want_pg:
            dw, db = dw.view(ctx.shapes[0]), db.view(ctx.shapes[1])
        return dx, dw if ctx.affine and ctx.needs_input_grad[1] else None, db if ctx.affine and ctx.needs_input_grad[2] else None, None


def layer_norm_4d(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], eps: float) -> Tensor:
    return _apply(LayerNorm4dFn, x, weight, bias, eps)


def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float, out_f32: bool = False) -> Tensor:
    """`out_f32`: an f32 input gives an f32 output — for a LayerNorm whose output is the RESIDUAL STREAM of the blocks behind it
    (nn.LayerNorm keeps its input's dtype under the reference's autocast; a bf16 copy is what a matrix product wants)."""
    if out_f32:
        return _apply(LayerNormFn, x, weight, bias, eps, True)
    return _apply(LayerNormFn, x, weight, bias, eps)


# ---------------------------------------------------------------------------------------------
# attention core (K3/K4)
# ---------------------------------------------------------------------------------------------


class PackedSelfAttentionFn(Function):
    """softmax(q k^T / sqrt(dh) [mask]) v on the packed projection output qkv [B, T, 3*D]
    (q | k | v along the last dim, head h = channels [h*64, (h+1)*64) of each third — the layout of
    reference attentions.py:214-216,180-185).  Output [B, T, D], heads already merged."""

    @staticmethod
    def forward(ctx: Any, qkv: Tensor, num_heads: int, keep_mask: Optional[Tensor], causal: bool,
                dropout_p: float = 0.0, head_dim: int = 64) -> Tensor:
        if qkv.dtype != bf16:
            qkv = ops.to_bf16(qkv.float().contiguous())
        if not qkv.is_contiguous():
            qkv = qkv.contiguous()
        d = qkv.shape[-1] // 3
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        ctx.drop = _take_attn_dropout(dropout_p, qkv.shape[0], num_heads, qkv.shape[1], qkv.shape[1])
        ctx.hd = {} if head_dim == 64 else dict(head_dim=head_dim)  # (any multiple of 8 up to 192: the `_dh` kernels)
        o, lse = ops.attn_fwd(q, k, v, num_heads, mask=keep_mask, causal=causal, **ctx.hd, **ctx.drop)
        ctx.save_for_backward(qkv, o, lse, keep_mask)
        ctx.num_heads, ctx.causal = num_heads, causal
        return o

    @staticmethod
    def backward(ctx: Any, d_o: Tensor):  # type: ignore
        qkv, o, lse, keep_mask = ctx.saved_tensors
        d = qkv.shape[-1] // 3
        if d_o.dtype != bf16:
            d_o = ops.to_bf16(d_o.float())
        d_o = d_o.contiguous()
        dqkv = torch.empty_like(qkv)
        ops.attn_bwd(
            qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], o, d_o, lse, ctx.num_heads,
            dq=dqkv[..., :d], dk=dqkv[..., d:2 * d], dv=dqkv[..., 2 * d:], mask=keep_mask,
            causal=ctx.causal, **ctx.hd, **ctx.drop,
        )
        return dqkv, None, None, None, None, None


class AttentionCoreFn(Function):
    """Same computation for separately projected q [B,Tq,D], k / v [B,Tk,D] (cross attention), any head_dim that is
    a multiple of 8 up to 192 (reference attentions.py:498-569 `CrossAttention` -> sdp_attn)."""

    @staticmethod
    def forward(ctx: Any, q: Tensor, k: Tensor, v: Tensor, num_heads: int, keep_mask: Optional[Tensor],
                causal: bool, head_dim: int = 64, dropout_p: float = 0.0) -> Tensor:
        q, k, v = (t if t.dtype == bf16 else ops.to_bf16(t.float().contiguous()) for t in (q, k, v))
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()  # equal shapes -> k and v share strides
        ctx.drop = _take_attn_dropout(dropout_p, q.shape[0], num_heads, q.shape[1], k.shape[1])
        o, lse = ops.attn_fwd(q, k, v, num_heads, mask=keep_mask, causal=causal, head_dim=head_dim, **ctx.drop)
        ctx.save_for_backward(q, k, v, o, lse, keep_mask)
        ctx.num_heads, ctx.causal, ctx.head_dim = num_heads, causal, head_dim
        return o

    @staticmethod
    def backward(ctx: Any, d_o: Tensor):  # type: ignore
        q, k, v, o, lse, keep_mask = ctx.saved_tensors
        if d_o.dtype != bf16:
            d_o = ops.to_bf16(d_o.float())
        d_o = d_o.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ops.attn_bwd(q, k, v, o, d_o, lse, ctx.num_heads, dq=dq, dk=dk, dv=dv, mask=keep_mask, causal=ctx.causal,
                     head_dim=ctx.head_dim, **ctx.drop)
        return dq, dk, dv, None, None, None, None, None


class AttentionWeightsFn(Function):
    """(output, weights) for separately projected q / k / v — the reference's slow path `require_weights=True` /
    `customize_sdp` (attentions.py:256-268).  The output is the fused kernels' (scale = 1 / `scaling`: the slow path is
    the one place where the reference honours `qk_scale`); the weights f32 [B, H, Tq, Tk] are rebuilt from the saved
    log-sum-exp (`cfhip_attn_probs`); a gradient on the weights reaches q and k through `cfhip_attn_probs_bwd`.
    `dropout_p` > 0 (training, attentions.py:263-264: `F.dropout(weights)` BEFORE `_weights_callback` and the product with
    v): the output comes from the dropout forms of the fused kernels and the returned weights are the softmax times the SAME
    Philox keep-mask (`cfhip_attn_dropout_mask` of the call's (seed, offset)) / (1 - p): what multiplied v, as in the reference."""

    @staticmethod
    def forward(ctx: Any, q: Tensor, k: Tensor, v: Tensor, num_heads: int, keep_mask: Optional[Tensor], causal: bool,
                head_dim: int, scale: float, dropout_p: float = 0.0):
        q, k, v = (t if t.dtype == bf16 else ops.to_bf16(t.float().contiguous()) for t in (q, k, v))
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        ctx.drop = _take_attn_dropout(dropout_p, q.shape[0], num_heads, q.shape[1], k.shape[1])
        o, lse = ops.attn_fwd(q, k, v, num_heads, mask=keep_mask, causal=causal, head_dim=head_dim, scale=scale, **ctx.drop)
        probs = ops.attn_probs(q, k, lse, num_heads, mask=keep_mask, causal=causal, head_dim=head_dim, scale=scale)
        if ctx.drop:
            probs.mul_(AttentionWeightsFn._drop_factor(ctx.drop, probs))
        ctx.save_for_backward(q, k, v, o, lse, keep_mask)
        ctx.num_heads, ctx.causal, ctx.head_dim, ctx.scale = num_heads, causal, head_dim, scale
        ctx.set_materialize_grads(False)
        return o, probs

    @staticmethod
    def _drop_factor(drop: dict, like: Tensor) -> Tensor:
        """keep-mask / (1 - p) of the call, f32 [B, H, Tq, Tk]; p is quantised to 1 / 256 by the kernels (ops.attn_dropout_p)"""
        b, h, tq, tk = like.shape
        keep = ops.attn_dropout_mask(b, h, tq, tk, drop["dropout_p"], drop["seed"], drop["offset"], device=like.device)
        return keep.to(f32).mul_(1.0 / (1.0 - ops.attn_dropout_p(drop["dropout_p"])))

    @staticmethod
    def backward(ctx: Any, d_o: Optional[Tensor], d_p: Optional[Tensor]):  # type: ignore
        q, k, v, o, lse, keep_mask = ctx.saved_tensors
        kw = dict(mask=keep_mask, causal=ctx.causal, head_dim=ctx.head_dim, scale=ctx.scale)
        dq = dk = dv = None
        if d_o is not None:
            d_o = (d_o if d_o.dtype == bf16 else ops.to_bf16(d_o.float())).contiguous()
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            ops.attn_bwd(q, k, v, o, d_o, lse, ctx.num_heads, dq=dq, dk=dk, dv=dv, **kw, **ctx.drop)
        if d_p is not None:
            d_p = d_p.float().contiguous()
            if ctx.drop:  # weights = softmax * keep / (1 - p): the gradient on the softmax is the gradient on the weights times the same factor
                d_p = d_p * AttentionWeightsFn._drop_factor(ctx.drop, d_p)
            dq2, dk2 = ops.attn_probs_bwd(q, k, lse, d_p, ctx.num_heads, **kw)
            dq = dq2 if dq is None else ops.add(dq, dq2)
            dk = dk2 if dk is None else ops.add(dk, dk2)
        return dq, dk, dv, None, None, None, None, None, None


def attention_with_weights(q: Tensor, k: Tensor, v: Tensor, num_heads: int, keep_mask: Optional[Tensor] = None,
                           causal: bool = False, head_dim: int = 64, scale: Optional[float] = None, dropout_p: float = 0.0):
    return _apply(AttentionWeightsFn, q, k, v, num_heads, keep_mask, causal, head_dim,
                                    1.0 / math.sqrt(float(head_dim)) if scale is None else float(scale), float(dropout_p))


def _take_attn_dropout(dropout_p: float, b: int, num_heads: int, tq: int, tk: int) -> dict:
    """kernel arguments of one attention call with dropout on the probabilities: draws its (seed, offset) from the
    process-wide Philox stream; the backward passes get the same pair and regenerate the mask"""
    if not 0.0 < dropout_p < 1.0:
        return {}
    seed, offset = ops.PhiloxState.take(ops.attn_dropout_blocks(b, num_heads, tq, tk))
    return dict(dropout_p=float(dropout_p), seed=seed, offset=offset)


def packed_self_attention(qkv: Tensor, num_heads: int, keep_mask: Optional[Tensor] = None,
                          causal: bool = False, dropout_p: float = 0.0, head_dim: int = 64) -> Tensor:
    return _apply(PackedSelfAttentionFn, qkv, num_heads, keep_mask, causal, dropout_p, head_dim)


def attention_core(q: Tensor, k: Tensor, v: Tensor, num_heads: int, keep_mask: Optional[Tensor] = None,
                   causal: bool = False, head_dim: int = 64, dropout_p: float = 0.0) -> Tensor:
    return _apply(AttentionCoreFn, q, k, v, num_heads, keep_mask, causal, head_dim, dropout_p)


# ---------------------------------------------------------------------------------------------
# small element-wise Functions used by the composed (non-fused) module paths
# ---------------------------------------------------------------------------------------------


class AddFn(Function):
    @staticmethod
    def forward(ctx: Any, a: Tensor, b: Tensor) -> Tensor:
        if a.dim() == 4 and (is_nhwc(a) or is_nhwc(b)):  # the residual adds of the UNet on NHWC rows (a layout mismatch: one transpose)
            a2, b2 = to_nhwc(a), to_nhwc(b)
            n, c, h, w = a2.shape
            return rows_to_nhwc(ops.add(nhwc_rows(a2), nhwc_rows(b2)), n, c, h, w)
        a2 = a if a.dtype == bf16 else ops.to_bf16(a.float().contiguous())
        b2 = b if b.dtype == bf16 else ops.to_bf16(b.float().contiguous())
        return ops.add(a2.contiguous(), b2.contiguous())

    @staticmethod
    def backward(ctx: Any, g: Tensor):  # type: ignore
        return g, g


class GeluFn(Function):
    @staticmethod
    def forward(ctx: Any, x: Tensor) -> Tensor:
        x2 = (x if x.dtype == bf16 else ops.to_bf16(x.float().contiguous())).contiguous()
        ctx.save_for_backward(x2)
        return ops.gelu_fwd(x2)

    @staticmethod
    def backward(ctx: Any, g: Tensor):  # type: ignore
        (x2,) = ctx.saved_tensors
        g2 = (g if g.dtype == bf16 else ops.to_bf16(g.float().contiguous())).contiguous()
        return ops.gelu_bwd(g2, x2)


def add(a: Tensor, b: Tensor) -> Tensor:
    return _apply(AddFn, a, b)


def gelu(x: Tensor) -> Tensor:
    return _apply(GeluFn, x)


# ---------------------------------------------------------------------------------------------
# ViT input stage (K8 as GEMM + K7 glue)
# ---------------------------------------------------------------------------------------------


class PatchTokensFn(Function):
    """VanillaPatchEmbed (stride == kernel Conv2d, reference high_level.py:172-188) + head-token
    concat + positional-encoding add (mixed_stacks/api.py:419-438,209-228) in three launches:
    im2row -> GEMM(+bias) -> assemble.  img [B,C,H,W] -> tokens bf16 [B, Np+1, D]."""

    @staticmethod
    def forward(ctx: Any, img: Tensor, conv_w: Tensor, conv_b: Optional[Tensor], head_token: Tensor,
                pos: Tensor) -> Tensor:
        b = img.shape[0]
        patch = conv_w.shape[-1]
        src = img if img.dtype in (f32, bf16) else img.float()
        rows = ops.im2row(src.contiguous(), patch)
        w16 = shadow_bf16(conv_w).view(conv_w.shape[0], -1)
        bias_f = None if conv_b is None else conv_b.detach().contiguous()
        patches = ops.gemm(rows, w16, bias=bias_f)
        x0 = ops.assemble_tokens_fwd(patches, head_token.detach().reshape(-1).contiguous(),
                                     pos.detach().reshape(-1).contiguous(), b)
        ctx.save_for_backward(rows)
        ctx.params = (conv_w, conv_b, head_token, pos)
        return x0

    @staticmethod
    def backward(ctx: Any, dx0: Tensor):  # type: ignore
        (rows,) = ctx.saved_tensors
        conv_w, conv_b, head_token, pos = ctx.params
        dx0 = (dx0 if dx0.dtype == bf16 else ops.to_bf16(dx0.float())).contiguous()
        d = dx0.shape[-1]
        direct = all(_is_direct(p) for p in (head_token, pos))
        g_head = g_pos = None
        if direct:
            # both written by one kernel: hand it the two destinations
            for prm in (head_token, pos):
                if prm.grad is None:
                    prm.grad = grad_buffer(prm, zero=True)
                    prm._cfhip_fresh = False
            acc_h = not getattr(head_token, "_cfhip_fresh", False)
            acc_p = not getattr(pos, "_cfhip_fresh", False)
            if acc_h != acc_p:  # keep one accumulate flag for the single launch
                for prm in (head_token, pos):
                    if getattr(prm, "_cfhip_fresh", False):
                        prm.grad.zero_()
                acc_h = acc_p = True
            dpatches = ops.assemble_tokens_bwd(dx0, head_token.grad.view(-1), pos.grad.view(-1), acc_h)
            for prm in (head_token, pos):
                prm._cfhip_fresh = False
                for cb in grad_ready_callbacks:
                    cb(prm)
        else:
            g_head = torch.empty((d,), dtype=f32, device=dx0.device)
            g_pos = torch.empty((dx0.shape[1] * d,), dtype=f32, device=dx0.device)
            dpatches = ops.assemble_tokens_bwd(dx0, g_head, g_pos, False)
            g_head, g_pos = g_head.view(head_token.shape), g_pos.view(pos.shape)
        w2 = conv_w.view(conv_w.shape[0], -1) if not _is_direct(conv_w) else conv_w
        gw, gb = None, None
        n, k, m = conv_w.shape[0], rows.shape[1], rows.shape[0]
        split = ops.pick_split_k(n, k, m)
        if conv_w.requires_grad:
            def dw_into(out: Tensor, acc: bool) -> None:
                ops.gemm(dpatches, rows, a_trans=True, b_trans=True, out=out.view(n, k), accumulate=acc,
                         split_k=split)
            if _is_direct(conv_w):
                write_param_grad(conv_w, dw_into)
            else:
                gw = torch.empty(conv_w.shape, dtype=f32, device=dx0.device)
                dw_into(gw, False)
        if conv_b is not None and conv_b.requires_grad:
            if _is_direct(conv_b):
                write_param_grad(conv_b, lambda out, acc: ops.colsum(dpatches, out=out.view(-1), accumulate=acc))
            else:
                gb = ops.colsum(dpatches).view(conv_b.shape)
        return None, gw, gb, g_head, g_pos


def patch_tokens(img: Tensor, conv_w: Tensor, conv_b: Optional[Tensor], head_token: Tensor,
                 pos: Tensor) -> Tensor:
    return _apply(PatchTokensFn, img, conv_w, conv_b, head_token, pos)


# ---------------------------------------------------------------------------------------------
# Conv2d general form (K8) = im2row + K1 GEMM; BatchNorm (K9/K10); LeakyReLU; global average pool
# ---------------------------------------------------------------------------------------------


# ---- NHWC activations (round 5) ------------------------------------------------------------------------------------------------
# The implicit-GEMM convolutions read and write NHWC ROWS ([B * H * W, C] bf16); modules exchange [B, C, H, W] tensors like the
# reference.  With `NHWC[0]` set (modules.UNetDiffuser.forward does it for the duration of a forward) a convolution hands its
# rows on as a channels_last VIEW — logical shape [B, C, H, W], strides (H W C, 1, W C, C), no copy — and every Function of the
# UNet path (GroupNorm, up-sampling, skip concatenation, residual add, the token <-> image hops of the SpatialTransformer)
# works on the rows behind such a view: the 318 NCHW <-> NHWC transposes of a 64^2 x 8 step are gone.  A Function that meets a
# layout it has no kernel for converts EXPLICITLY with the transpose kernel (`to_nchw` / `to_nhwc`) — never through
# `.contiguous()`, which would be a silent ATen copy.  CFHIP_UNET_NHWC=0 keeps the round-4 NCHW hand-over.


class _ThreadFlag:
    """`flag[0]` per thread: a forward on another thread (EMA evaluation, a second model) keeps its own setting."""

    def __init__(self) -> None:
        self._local = threading.local()

    def __getitem__(self, i: int) -> bool:
        return getattr(self._local, "value", False)

    def __setitem__(self, i: int, value: Any) -> None:
        self._local.value = bool(value)


NHWC = _ThreadFlag()
NHWC_ENABLED = os.environ.get("CFHIP_UNET_NHWC", "1") != "0"


def is_nhwc(x: Tensor) -> bool:
    if x.dim() != 4:
        return False
    b, c, h, w = x.shape
    return c > 1 and h * w > 1 and x.stride() == (h * w * c, 1, w * c, c)


def nhwc_rows(x: Tensor) -> Tensor:
    """the [B * H * W, C] rows behind a channels_last view (no copy)"""
    b, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(b * h * w, c)


def rows_to_nhwc(rows: Tensor, b: int, c: int, h: int, w: int) -> Tensor:
    """[B * H * W, C] rows as a [B, C, H, W] channels_last view (no copy)"""
    return rows.view(b, h, w, c).permute(0, 3, 1, 2)


def to_nchw(x: Tensor) -> Tensor:
    """dense NCHW of a 4-D activation: the transpose kernel for a channels_last view, `.contiguous()` otherwise (no-op when dense)"""
    if not is_nhwc(x):
        return x.contiguous()
    b, c, h, w = x.shape
    return ops.transpose_batched(nhwc_rows(x).view(b, h * w, c)).view(b, c, h, w)


def to_nhwc(x: Tensor) -> Tensor:
    """bf16 channels_last view of a 4-D activation (the transpose kernel when it arrives NCHW)"""
    if is_nhwc(x) and x.dtype == bf16:
        return x
    if x.dtype not in (bf16, f32):
        x = x.float()
    x = to_nchw(x)
    b, c, h, w = x.shape
    return rows_to_nhwc(ops.transpose_batched(x.view(b, c, h * w)).view(b * h * w, c), b, c, h, w)


def as_bf16_act(t: Tensor) -> Tensor:
    """a 4-D activation / gradient as bf16 in the layout it arrived in (NHWC rows stay NHWC rows)"""
    if t.dtype == bf16:
        return t
    t = t.float()
    if is_nhwc(t):
        b, c, h, w = t.shape
        return rows_to_nhwc(ops.to_bf16(nhwc_rows(t)), b, c, h, w)
    return ops.to_bf16(t.contiguous())


def _conv_weight_rows(weight: Tensor, kp: int) -> Tensor:
    """bf16 [Cout, Kp] view / zero-padded copy of the [Cout, Cin, kh, kw] weight (k = (c, ky, kx))."""
    cout = weight.shape[0]
    k = weight.numel() // cout
    w16 = shadow_bf16(weight).view(cout, k)
    if kp == k:
        return w16
    wp = torch.zeros((cout, kp), dtype=bf16, device=weight.device)
    wp[:, :k] = w16  # first layer only (K = Cin*kh*kw not a multiple of 8): a few KB
    return wp


# 3x3 / stride 1 / pad 1 convolutions with Cin % 32 == 0 (every ResBlock / up-sampling conv of the UNet) skip the im2row
# matrix: cfhip_conv3x3_nhwc_bf16 gathers the taps inside the GEMM's K loop.  UNet 64^2 x 8 step: im2row + row2im were
# 38 ms of 140 (profiles/r01/prof_unet_summary.txt).  False: every convolution takes the im2row path (A/B, tests).
IMPLICIT_CONV = True


# Where the rotated 3x3 filters of the input-gradient convolution are packed.  2 (default): in the FORWARD of the layer, on the
# caller's stream — the UNet's forward is issue-bound (its queue drains faster than the host fills it) and its backward is
# bound by the queue, so the 35 us per layer cost nothing there and 1.65 ms per step here: 64^2 x 8 step 62.4 -> 61.8 ms
# (three alternating pairs, profiles/r04/unet_pack_ahead_ab.txt; +1 GB of packed filters alive until backward).  1: on the
# side lane beside the forward (fork + event per layer cost the host more than the queue gains: 62.3 vs 61.5).  0: in backward.
PACK_AHEAD = 2  # (A/B closed in round 4: profiles/r04/unet_pack_ahead_ab.txt; tests set the variable)


def _implicit_ok(cin: int, cout: int, kh: int, kw: int, stride: int, pad: int, dil: int, h: int, w: int) -> bool:
    return (IMPLICIT_CONV and (kh, kw, stride, pad, dil) == (3, 3, 1, 1, 1) and cin % 32 == 0 and cout % 8 == 0
            and h < 65536 and w < 65536)


# A 3x3 / stride 1 / pad 1 convolution onto a FEW channels that are not a multiple of 8 (the UNet's 3-channel head at full
# resolution) runs the implicit route on filters padded with zeros to 32 output channels: the im2row route materialises the
# [B H W, 9 Cin] matrix (189 MB at 64^2 x 8 x 320) in the forward, again for the weight gradient, and once more as the gradient
# matrix that row2im folds back; the padded implicit convolution reads the 21 MB activation.  32, not 8: the input-gradient
# convolution takes dY as its input and wants a multiple of 32 channels there.
THIN_HEAD_IMPLICIT = True  # (A/B closed in round 5: profiles/r05/conv_thin_head_ab.txt)
THIN_HEAD_PAD = 32


def _thin_head_ok(cin: int, cout: int, kh: int, kw: int, stride: int, pad: int, dil: int, h: int, w: int) -> bool:
    return (THIN_HEAD_IMPLICIT and cout % 8 != 0 and cout < THIN_HEAD_PAD
            and _implicit_ok(cin, THIN_HEAD_PAD, kh, kw, stride, pad, dil, h, w))


# ---- the filter matrices of a model's 3x3 convolutions packed in one launch at the top of its forward (round 6, late) -------------------
# Conv2dFn packs its filters into the two layouts of the implicit-GEMM kernel (forward; rotated for dX) on every call: the weights change
# every step.  For the zoo UNet that is 96 launches of 12-15 us on the critical queue.  `prepack_convs` does all of them in two launches
# (cfhip_conv3x3_pack_filters_grouped) into buffers that live with the parameters; Conv2dFn.forward picks its pair up with `pack_pre_lookup`
# while the scope of that forward lasts.  The buffers of a step are overwritten by the next forward's launch, i.e. after the backward that
# read them was issued on the same stream.
PACK_GROUPED = True  # (A/B: tools set the attribute)


class _PackPre(threading.local):
    def __init__(self) -> None:
        self.packed: dict = {}


_PACK_PRE = _PackPre()


def prepack_convs(convs: Sequence[Any]) -> bool:
    """pack the filters of `convs` (modules with a [Cout, Cin, 3, 3] `weight` that reach Conv2dFn's implicit route untransformed) now"""
    pack_pre_clear()
    if not PACK_GROUPED or not convs or not convs[0].weight.is_cuda:
        return False
    items, packed = [], {}
    for conv in convs:
        w = conv.weight
        cout, cin = w.shape[0], w.shape[1]
        w16 = shadow_bf16(w).view(cout, cin, 3, 3)
        ent = getattr(w, "_cfhip_packed", None)  # (the buffers live on the parameter object: a deep copy of the model starts without them)
        if ent is None or ent[0].device != w16.device:
            ent = (torch.empty((cout, 9 * cin), dtype=bf16, device=w16.device),
                   torch.empty((cin, 9 * cout), dtype=bf16, device=w16.device) if cout % 32 == 0 else None)
            w._cfhip_packed = ent
        items.append((w16, ent[0], False))
        if ent[1] is not None:
            items.append((w16, ent[1], True))
        packed[id(w)] = ent
    ops.conv3x3_pack_grouped(items)
    _PACK_PRE.packed = packed
    return True


def pack_pre_lookup(weight: Tensor):
    """(forward filter matrix, rotated one or None) of `weight` out of this forward's grouped launch, else None"""
    return _PACK_PRE.packed.get(id(weight))


def pack_pre_clear() -> None:
    _PACK_PRE.packed = {}


class Conv2dFn(Function):
    """Replaces F.conv2d reached from Conv2d.forward (reference convs/basic.py:160-177), groups = 1:
    NCHW in (f32 / bf16), NCHW bf16 out.  Two routes: implicit GEMM on an NHWC copy (3x3 / stride 1 / pad 1,
    Cin % 32 == 0: forward always, input gradient when Cout % 32 == 0 too) or im2row + GEMM (everything else, and
    every weight gradient)."""

    @staticmethod
    def forward(ctx: Any, x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int, pad: int,
                dil: int) -> Tensor:
        if x.dtype not in (bf16, f32):
            x = x.float()
        nhwc_in = is_nhwc(x) and x.dtype == bf16  # the producer handed its NHWC rows on: no transpose on the way in
        ctx.nhwc_in = nhwc_in
        if not nhwc_in:
            x = to_nchw(x)
        b, cin, h, w = x.shape
        cout, _, kh, kw = weight.shape
        ho, wo = ops.conv_out_hw(h, w, kh, kw, stride, pad, dil)
        bias_f = None if bias is None else bias.detach().reshape(-1).contiguous()
        ctx.weight, ctx.bias = weight, bias
        ctx.geom = (kh, kw, stride, pad, dil, ho, wo)
        ctx.implicit = _implicit_ok(cin, cout, kh, kw, stride, pad, dil, h, w)
        thin = not ctx.implicit and _thin_head_ok(cin, cout, kh, kw, stride, pad, dil, h, w)
        nhwc_out = NHWC[0] and cout % 8 == 0  # hand the output rows on as a channels_last view
        if ctx.implicit or thin:
            ctx.implicit = True
            x_rows = nhwc_rows(x) if nhwc_in else ops.transpose_batched(x.view(b, cin, h * w)).view(b * h * w, cin)  # NHWC bf16
            w16 = shadow_bf16(weight).view(cout, cin, 3, 3)
            cop = cout
            if thin:  # zero filters up to THIN_HEAD_PAD output channels (see _thin_head_ok); backward reads the padded copy
                cop = THIN_HEAD_PAD
                w16p = torch.zeros((cop, cin, 3, 3), dtype=bf16, device=x.device)
                w16p[:cout] = w16
                w16 = w16p
                if bias_f is not None:
                    bpad = torch.zeros((cop,), dtype=f32, device=x.device)
                    bpad[:cout] = bias_f
                    bias_f = bpad
            pre = None if thin else pack_pre_lookup(weight)  # both filter matrices out of the model's one launch (prepack_convs)
            wk = pre[0] if pre is not None else ops.conv3x3_pack_filters(w16, False)  # k = (ky, kx, c): a K-step = 32 channels of a tap
            y_rows = ops.conv3x3_nhwc(x_rows, wk, bias_f, b, h, w)
            ctx.save_for_backward(x_rows, w16)  # the NHWC bf16 copy serves the weight gradient; x itself is not kept
            ctx.xshape = (b, cin, h, w)
            # the input gradient wants the filters rotated by 180 degrees with the channels swapped: packed HERE, on the side
            # lane beside the forward convolution, instead of on the backward's critical queue (47 launches, 1.65 ms of the
            # 64^2 x 8 UNet step).  The weights' current bf16 shadow does not change before this step's backward has run.
            ctx.wr = ctx.wr_event = None
            if pre is not None and pre[1] is not None and ctx.needs_input_grad[0]:
                ctx.wr = pre[1]
            elif PACK_AHEAD == 2 and cop % 32 == 0 and ctx.needs_input_grad[0]:
                # on the caller's stream: the forward of the UNet is issue-bound (the queue drains faster than the host fills
                # it), the backward is bound by its queue — the pack costs nothing here and 35 us per layer there
                ctx.wr = ops.conv3x3_pack_filters(w16, True)
            elif PACK_AHEAD == 1 and cop % 32 == 0 and ctx.needs_input_grad[0]:
                side = SideStream.fork(0)
                if side is not None:
                    wr = torch.empty((cin, 9 * cop), dtype=bf16, device=x.device)  # (allocated on the caller's stream: freed there)
                    with on_stream(side):
                        ops.conv3x3_pack_filters(w16, True, out=wr)
                    ev = torch.cuda.Event()
                    ev.record(side)
                    ctx.wr, ctx.wr_event = wr, ev
            if nhwc_out:
                return rows_to_nhwc(y_rows, b, cout, h, w)
            y = ops.transpose_batched(y_rows.view(b, h * w, cop)).view(b, cop, h, w)
            return y[:, :cout].contiguous() if cop != cout else y
        # 1x1 / stride 1 (the skip connections of the UNet's residual blocks): the im2row matrix IS the NHWC copy of x — one
        # batched transpose instead of the gather kernel, kept for the weight gradient (the im2row route recomputes it), and
        # dX comes back through a transpose instead of row2im (UNet 64^2 x 8: 38 im2row + 18 row2im launches, 3.7 ms)
        ctx.pointwise = (kh, kw, stride, pad, dil) == (1, 1, 1, 0, 1) and cin % 8 == 0
        if ctx.pointwise:
            rows = nhwc_rows(x) if nhwc_in else ops.transpose_batched(x.view(b, cin, h * w)).view(b * h * w, cin)
        else:
            if nhwc_in:  # strided / odd-shaped convolutions gather from NCHW (3 down-sampling layers per UNet): one transpose
                x = to_nchw(x)
            rows = ops.conv_im2row(x, kh, kw, stride, pad, dil)
        wp = _conv_weight_rows(weight, rows.shape[1])
        # Cout that is not a multiple of 8 (the 3-channel head of the UNet) would fall off the MFMA path onto the
        # one-thread-per-output kernel (3.6 ms per launch at 64^2 x 8): pad the output channels with zero filters
        cp = (cout + 7) // 8 * 8
        if cp != cout:
            wpad = torch.zeros((cp, wp.shape[1]), dtype=bf16, device=x.device)
            wpad[:cout] = wp
            wp = wpad
            if bias_f is not None:
                bpad = torch.zeros((cp,), dtype=f32, device=x.device)
                bpad[:cout] = bias_f
                bias_f = bpad
        y_rows = ops.gemm(rows, wp, bias=bias_f)  # [B*Ho*Wo, Cp] bf16
        if nhwc_out and cp == cout:
            y = rows_to_nhwc(y_rows, b, cout, ho, wo)
        else:
            y = ops.transpose_batched(y_rows.view(b, ho * wo, cp)).view(b, cp, ho, wo)
        if cp != cout:
            y = y[:, :cout].contiguous()
        ctx.save_for_backward(rows if ctx.pointwise else x, wp)
        ctx.xshape = (b, cin, h, w)
        return y

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        x, wp = ctx.saved_tensors  # implicit route: x is the NHWC row matrix [B*H*W, Cin]
        weight, bias = ctx.weight, ctx.bias
        kh, kw, stride, pad, dil, ho, wo = ctx.geom
        b, cin, h, w = ctx.xshape
        cout = weight.shape[0]
        k = cin * kh * kw
        dy = as_bf16_act(dy)
        # implicit forward: wp is the [Cout, Cin, 3, 3] bf16 shadow, the dW GEMM still reads an im2row matrix
        cp = wp.shape[0]  # output channels incl. the zero filters padding Cout (to 8 on the im2row route, to 32 for a thin implicit head)
        if cp != cout:
            dyp = torch.zeros((b, cp, ho * wo), dtype=bf16, device=dy.device)
            dyp[:, :cout] = to_nchw(dy).reshape(b, cout, ho * wo)
            dy = dyp
        if is_nhwc(dy):
            dy_rows = nhwc_rows(dy)  # the consumer's gradient arrives as NHWC rows: no transpose
        else:
            dy_rows = ops.transpose_batched(dy.contiguous().view(b, cp, ho * wo)).view(b * ho * wo, cp)
        nhwc_dx = getattr(ctx, "nhwc_in", False)  # the input came as NHWC rows: its gradient goes back the same way
        wgrad_implicit = ctx.implicit and ops.conv3x3_wgrad_ok(b, h, w)

        def param_grads(x: Tensor = x) -> Tuple[Optional[Tensor], Optional[Tensor]]:
            gw = gb = None
            if ctx.implicit and not wgrad_implicit and weight.requires_grad:
                x = ops.transpose_batched(x.view(b, h * w, cin)).view(b, cin, h, w)  # back to NCHW for the im2row route
            if weight.requires_grad and wgrad_implicit:
                split = ops.pick_split_k(cp, 9 * cin, b * h * w)

                def dw_implicit(out: Tensor, acc: bool) -> None:
                    if cp == cout:
                        ops.conv3x3_wgrad_nhwc(dy_rows, x, b, h, w, split, out=out.view(cout, cin, 3, 3), accumulate=acc)
                        return
                    tmp = ops.conv3x3_wgrad_nhwc(dy_rows, x, b, h, w, split)[:cout]  # (the zero filters' rows are dropped)
                    if acc:
                        out.view(cout, cin, 3, 3).add_(tmp)
                    else:
                        out.view(cout, cin, 3, 3).copy_(tmp)

                if _is_direct(weight):
                    write_param_grad(weight, dw_implicit)
                else:
                    gw = torch.empty(weight.shape, dtype=f32, device=dy.device)
                    dw_implicit(gw, False)
            elif weight.requires_grad:
                # (im2row is recomputed: k^2 times the input, not kept; a 1x1 convolution saved its NHWC rows instead)
                rows = x if getattr(ctx, "pointwise", False) else ops.conv_im2row(x, kh, kw, stride, pad, dil)
                m = rows.shape[0]
                kp = rows.shape[1]
                split = ops.pick_split_k(cp, kp, m)

                def dw_into(out: Tensor, acc: bool) -> None:
                    if kp == k and cp == cout:
                        ops.gemm(dy_rows, rows, a_trans=True, b_trans=True, out=out.view(cout, k), accumulate=acc,
                                 split_k=split)
                    else:
                        tmp = ops.gemm(dy_rows, rows, a_trans=True, b_trans=True, out_dtype=f32, split_k=split)
                        if acc:
                            out.view(cout, k).add_(tmp[:cout, :k])
                        else:
                            out.view(cout, k).copy_(tmp[:cout, :k])

                if _is_direct(weight):
                    write_param_grad(weight, dw_into)
                else:
                    gw = torch.empty(weight.shape, dtype=f32, device=dy.device)
                    dw_into(gw, False)
            if bias is not None and bias.requires_grad:
                def db_into(out: Tensor, acc: bool) -> None:
                    if cp == cout:
                        ops.colsum(dy_rows, out=out.view(-1), accumulate=acc)
                    else:
                        tmp = ops.colsum(dy_rows)[:cout]
                        if acc:
                            out.view(-1).add_(tmp)
                        else:
                            out.view(-1).copy_(tmp)

                if _is_direct(bias):
                    write_param_grad(bias, db_into)
                else:
                    gb = torch.empty(bias.shape, dtype=f32, device=dy.device)
                    db_into(gb, False)
            return gw, gb

        if _all_direct(weight, bias):
            # nothing goes back through autograd: the weight / bias gradients run on the side stream beside dX
            SideStream.run(param_grads, (dy_rows, x))
            gw = gb = None
        else:
            gw, gb = param_grads()
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.implicit and cp % 32 == 0:
                # dX = conv3x3(dY, filters rotated by 180 degrees, channels swapped): k = (ky, kx, co)
                wr = getattr(ctx, "wr", None)
                if wr is not None:
                    if ctx.wr_event is not None:
                        cur_stream().wait_event(ctx.wr_event)
                else:
                    wr = ops.conv3x3_pack_filters(wp, True)
                dx_rows = ops.conv3x3_nhwc(dy_rows, wr, None, b, h, w)
                dx = rows_to_nhwc(dx_rows, b, cin, h, w) if nhwc_dx else ops.transpose_batched(dx_rows.view(b, h * w, cin)).view(b, cin, h, w)
            else:
                w2 = wp.reshape(cp, k) if ctx.implicit else wp
                drows = ops.gemm(dy_rows, w2, b_trans=True)  # [M, Kp] bf16
                if getattr(ctx, "pointwise", False):
                    dx = rows_to_nhwc(drows, b, cin, h, w) if nhwc_dx else ops.transpose_batched(drows.view(b, h * w, cin)).view(b, cin, h, w)
                else:
                    dx = ops.conv_row2im(drows, (b, cin, h, w), kh, kw, stride, pad, dil)
                    if nhwc_dx:
                        dx = to_nhwc(dx)
        return dx, gw, gb, None, None, None


class GroupedConv2dFn(Function):
    """F.conv2d with groups > 1 (reference convs/basic.py:160-177; depthwise = groups == Cin): the direct kernels of
    csrc/conv_grouped.hip — an option outside the benchmark configurations, correct and deterministic, not tuned."""

    @staticmethod
    def forward(ctx: Any, x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int, pad: int, dil: int, groups: int) -> Tensor:
        x2 = to_nchw(as_bf16_act(x))
        w16 = shadow_bf16(weight).view(weight.shape)
        bias_f = None if bias is None else bias.detach().reshape(-1).contiguous()
        ctx.save_for_backward(x2, w16)
        ctx.weight, ctx.bias, ctx.geom = weight, bias, (stride, pad, dil, groups)
        return ops.conv2d_grouped_fwd(x2, w16, bias_f, stride, pad, dil, groups)

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        x2, w16 = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        stride, pad, dil, groups = ctx.geom
        dy = to_nchw(as_bf16_act(dy))
        want_b = bias is not None and bias.requires_grad
        gw = gb = None
        if weight.requires_grad or want_b:
            direct_w, direct_b = _is_direct(weight), (want_b and _is_direct(bias))
            dw = db = None
            acc_w = acc_b = False
            if weight.requires_grad and direct_w:
                if weight.grad is None:
                    weight.grad = grad_buffer(weight)
                    weight._cfhip_fresh = True
                dw, acc_w = weight.grad, not getattr(weight, "_cfhip_fresh", False)
            else:
                dw = gw = torch.empty(weight.shape, dtype=f32, device=dy.device)  # (also the scratch of a bias-only request)
            if want_b and direct_b:
                if bias.grad is None:
                    bias.grad = grad_buffer(bias)
                    bias._cfhip_fresh = True
                db, acc_b = bias.grad.view(-1), not getattr(bias, "_cfhip_fresh", False)
            elif want_b:
                db = gb = torch.empty(bias.shape, dtype=f32, device=dy.device)
            ops.conv2d_grouped_bwd_weight(dy, x2, dw, acc_w, db, acc_b, stride, pad, dil, groups)
            for prm, direct in ((weight, weight.requires_grad and direct_w), (bias, direct_b)):
                if direct:
                    prm._cfhip_fresh = False
                    notify_grad_ready(prm)
            if not weight.requires_grad:
                gw = None
        dx = ops.conv2d_grouped_bwd_input(dy, w16, x2.shape, stride, pad, dil, groups) if ctx.needs_input_grad[0] else None
        return dx, gw, gb, None, None, None, None


class ConvTranspose2dFn(Function):
    """F.conv_transpose2d(x, wt, None, stride, padding, dilation=dilation), groups 1, output_padding 0 — what the reference's
    `Conv2d.forward(transpose=True)` reaches (convs/basic.py:160-177).  A transposed convolution IS the input-gradient map of the
    convolution with the same weight tensor read as [C_conv_out = Cin, C_conv_in = Cout, k, k], so the three products are the ones
    `Conv2dFn` already issues with the roles of forward and backward swapped:
      forward   y  = row2im(x_rows wt2)                       (Conv2dFn's dX route: GEMM (nn) + gather of the overlapping windows)
      backward  dx = im2row(dy) wt2^T   (Conv2dFn's forward), dwt2 = x_rows^T im2row(dy)   (its weight gradient, tn GEMM)
    x [B, Cin, h, w] -> bf16 [B, Cout, (h - 1) s - 2 p + d (k - 1) + 1, ...]; wt [Cin, Cout, k, k]."""

    @staticmethod
    def forward(ctx: Any, x: Tensor, wt: Tensor, stride: int, pad: int, dil: int) -> Tensor:
        if x.dtype not in (bf16, f32):
            x = x.float()
        x = to_nchw(x)
        b, cin, h, w = x.shape
        cin_w, cout, kh, kw = wt.shape
        if cin_w != cin:
            raise ValueError(f"conv_transpose2d: input has {cin} channels, the weight expects {cin_w}")
        ho = (h - 1) * stride - 2 * pad + dil * (kh - 1) + 1
        wo = (w - 1) * stride - 2 * pad + dil * (kw - 1) + 1
        if ho <= 0 or wo <= 0:
            raise ValueError("conv_transpose2d: empty output")
        k = cout * kh * kw
        kp = (k + 7) // 8 * 8
        x_rows = ops.transpose_batched(x.view(b, cin, h * w)).view(b * h * w, cin)  # NHWC rows, bf16
        w2 = _conv_weight_rows(wt, kp)                                             # [Cin, Kp] bf16
        drows = ops.gemm(x_rows, w2, b_trans=True)                                 # [B h w, Kp]
        y = ops.conv_row2im(drows, (b, cout, ho, wo), kh, kw, stride, pad, dil)
        ctx.save_for_backward(x_rows, w2)
        ctx.wt = wt
        ctx.geom = (b, cin, h, w, cout, kh, kw, stride, pad, dil, k, kp)
        return y

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        x_rows, w2 = ctx.saved_tensors
        wt = ctx.wt
        b, cin, h, w, cout, kh, kw, stride, pad, dil, k, kp = ctx.geom
        rows = ops.conv_im2row(to_nchw(as_bf16_act(dy)), kh, kw, stride, pad, dil)  # [B h w, Kp]: the convolution's view of dy
        dx = gw = None
        if ctx.needs_input_grad[0]:
            if cin % 8 == 0:
                dx_rows = ops.gemm(rows, w2)  # [B h w, Cin]
            else:  # (output columns that are no multiple of 8 would leave the MFMA path: pad with zero filters, as Conv2dFn does)
                cp = (cin + 7) // 8 * 8
                wpad = torch.zeros((cp, kp), dtype=bf16, device=dy.device)
                wpad[:cin] = w2
                dx_rows = ops.gemm(rows, wpad)[:, :cin].contiguous()
            dx = ops.transpose_batched(dx_rows.view(b, h * w, cin)).view(b, cin, h, w)
        if wt.requires_grad:
            if cin % 8 == 0:
                g2 = ops.gemm(x_rows, rows, a_trans=True, b_trans=True, out_dtype=f32, split_k=ops.pick_split_k(cin, kp, rows.shape[0]))
            else:
                g2 = ops.gemm(x_rows, rows, a_trans=True, b_trans=True, out_dtype=f32)
            gw = g2[:, :k].reshape(wt.shape)
        return dx, gw, None, None, None


def conv_transpose2d(x: Tensor, wt: Tensor, stride: int, pad: int, dil: int = 1) -> Tensor:
    return _apply(ConvTranspose2dFn, x, wt, stride, pad, dil)


def conv2d(x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int, pad: int, dil: int = 1, groups: int = 1) -> Tensor:
    if groups != 1:
        return _apply(GroupedConv2dFn, x, weight, bias, stride, pad, dil, groups)
    return _apply(Conv2dFn, x, weight, bias, stride, pad, dil)


class BatchNormFn(Function):
    """Replaces nn.BatchNorm1d/2d.forward (reference norms.py:20-27,90-93) incl. the running-statistics update."""

    @staticmethod
    def forward(ctx: Any, x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor],
                running_mean: Optional[Tensor], running_var: Optional[Tensor], eps: float, momentum: float,
                training: bool) -> Tensor:
        if x.dtype not in (bf16, f32):
            x = x.float()
        x = to_nchw(x)
        gamma = None if weight is None else weight.detach().contiguous()
        beta = None if bias is None else bias.detach().contiguous()
        y, mean, rstd = ops.batchnorm_fwd(x, gamma, beta, running_mean, running_var, eps, momentum, training)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.weight, ctx.bias, ctx.training = weight, bias, training
        return y

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        x, gamma, mean, rstd = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        dy = to_nchw(as_bf16_act(dy)) if dy.dim() == 4 else (dy if dy.dtype == bf16 else ops.to_bf16(dy.float().contiguous()))
        dx, dg, db = ops.batchnorm_bwd(dy, x, gamma, mean, rstd, training=ctx.training,
                                       want_dx=ctx.needs_input_grad[0])
        gw = gb = None
        for prm, g, which in ((weight, dg, 0), (bias, db, 1)):
            if prm is None or not prm.requires_grad:
                continue
            if _is_direct(prm):
                write_param_grad(prm, lambda out, acc, g=g: out.add_(g.view(out.shape)) if acc else out.copy_(g.view(out.shape)))
            elif which == 0:
                gw = g.view(prm.shape)
            else:
                gb = g.view(prm.shape)
        return dx, gw, gb, None, None, None, None, None


def batch_norm(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], running_mean: Optional[Tensor],
               running_var: Optional[Tensor], eps: float, momentum: float, training: bool) -> Tensor:
    return _apply(BatchNormFn, x, weight, bias, running_mean, running_var, eps, momentum, training)


class LeakyReLUFn(Function):
    """nn.LeakyReLU(slope) / nn.ReLU (slope 0) (reference activations.py:35-43)."""

    @staticmethod
    def forward(ctx: Any, x: Tensor, slope: float) -> Tensor:
        if x.dtype != bf16:
            x = ops.to_bf16(x.float().contiguous())
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.slope = slope
        return ops.leaky_relu_fwd(x, slope)

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        (x,) = ctx.saved_tensors
        if dy.dtype != bf16:
            dy = ops.to_bf16(dy.float().contiguous())
        return ops.leaky_relu_bwd(dy.contiguous(), x, ctx.slope), None


def leaky_relu(x: Tensor, slope: float) -> Tensor:
    return _apply(LeakyReLUFn, x, slope)


class GlobalAvgPoolFn(Function):
    """nn.AdaptiveAvgPool2d((1, 1)) + squeeze (reference cv/encoder/vanilla.py:144,155-158): [B,C,H,W] -> [B,C]."""

    @staticmethod
    def forward(ctx: Any, x: Tensor) -> Tensor:
        if x.dtype != bf16:
            x = ops.to_bf16(x.float().contiguous())
        ctx.shape = tuple(x.shape)
        return ops.avgpool_fwd(x.contiguous())

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        if dy.dtype != bf16:
            dy = ops.to_bf16(dy.float().contiguous())
        return ops.avgpool_bwd(dy.contiguous(), ctx.shape)


def global_avg_pool(x: Tensor) -> Tensor:
    return _apply(GlobalAvgPoolFn, x)


class FocalLossFn(Function):
    """mean focal loss (reference losses/basic.py:170-206) on f32 logits with int64 labels."""

    @staticmethod
    def forward(ctx: Any, logits: Tensor, labels: Tensor, gamma: float, eps: float) -> Tensor:
        b = logits.shape[0]
        loss, dlogits = ops.softmax_focal(logits.float().contiguous(), labels, 1.0 / b, gamma=gamma, eps=eps)
        ctx.save_for_backward(dlogits)
        return loss[0] / b

    @staticmethod
    def backward(ctx: Any, g: Tensor):  # type: ignore
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None, None, None


def focal_loss(logits: Tensor, labels: Tensor, gamma: float = 2.0, eps: float = 1.0e-6) -> Tensor:
    return _apply(FocalLossFn, logits, labels, gamma, eps)


# ---------------------------------------------------------------------------------------------
# CLIP text tower: embedding lookup (+ positional add), row gather (EOT pooling), L2 normalisation
# ---------------------------------------------------------------------------------------------


class EmbeddingFn(Function):
    """nn.Embedding lookup (reference multimodal/clip.py:160-164,238) optionally fused with the learned positional
    add of `PositionalEncoding` (mixed_stacks/api.py:209-228): out f32 [..., D]."""

    @staticmethod
    def forward(ctx: Any, indices: Tensor, weight: Tensor, pos: Optional[Tensor], padding_idx: int) -> Tensor:
        t = indices.shape[-1]
        posd = None if pos is None else pos.detach().reshape(-1, pos.shape[-1])[:t].contiguous()
        out = ops.embedding_fwd(weight.detach().contiguous(), indices, posd, period=t if pos is not None else 0)
        ctx.save_for_backward(indices)
        ctx.weight, ctx.pos, ctx.padding_idx, ctx.t = weight, pos, padding_idx, t
        return out

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        (indices,) = ctx.saved_tensors
        weight, pos = ctx.weight, ctx.pos
        gw = gp = None
        if weight.requires_grad:
            def scatter(out: Tensor, acc: bool) -> None:
                if not acc:
                    out.zero_()
                ops.embedding_bwd(dy, indices, out, ctx.padding_idx)

            if _is_direct(weight):
                write_param_grad(weight, scatter)
            else:
                gw = torch.zeros(weight.shape, dtype=f32, device=dy.device)
                ops.embedding_bwd(dy, indices, gw, ctx.padding_idx)
        if pos is not None and pos.requires_grad:
            d = dy.shape[-1]
            dy2 = dy if dy.dtype == bf16 else ops.to_bf16(dy.float().contiguous())
            g = ops.colsum(dy2.reshape(-1, ctx.t * d))  # sum over the batch of [B, T*D]
            gp = torch.zeros(pos.shape, dtype=f32, device=dy.device)
            gp.view(-1, d)[:ctx.t] = g.view(ctx.t, d)
        return None, gw, gp, None


def embedding(indices: Tensor, weight: Tensor, pos: Optional[Tensor] = None, padding_idx: int = -1) -> Tensor:
    return _apply(EmbeddingFn, indices, weight, pos, padding_idx)


class GatherRowsFn(Function):
    """x[arange(B), index] on the [B, T, D] f32 stream (reference multimodal/clip.py:247-249, EOT pooling)."""

    @staticmethod
    def forward(ctx: Any, x: Tensor, index: Tensor) -> Tensor:
        b, t, d = x.shape
        x2 = x.float().contiguous().view(b * t, d) if x.dtype != f32 else x.contiguous().view(b * t, d)
        flat = torch.arange(b, device=x.device, dtype=torch.int64) * t + index.reshape(-1)
        ctx.save_for_backward(flat)
        ctx.shape = (b, t, d)
        return ops.embedding_fwd(x2, flat)

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        (flat,) = ctx.saved_tensors
        b, t, d = ctx.shape
        dx = torch.zeros((b * t, d), dtype=f32, device=dy.device)
        ops.embedding_bwd(dy, flat, dx)
        return dx.view(b, t, d), None


def gather_rows(x: Tensor, index: Tensor) -> Tensor:
    return _apply(GatherRowsFn, x, index)


class L2NormalizeFn(Function):
    """cftool.array.l2_normalize (carefree-toolkit, not vendored): x / ||x||_2 over the last dim, no epsilon."""

    @staticmethod
    def forward(ctx: Any, x: Tensor) -> Tensor:
        y, inv = ops.l2norm_fwd(x.float().contiguous())
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        y, inv = ctx.saved_tensors
        return ops.l2norm_bwd(dy.float().contiguous(), y, inv)


def l2_normalize(x: Tensor) -> Tensor:
    return _apply(L2NormalizeFn, x)


# ---------------------------------------------------------------------------------------------
# UNet residual block pieces (reference convs/residual.py:86-253)
# ---------------------------------------------------------------------------------------------


class GroupNormFn(Function):
    """y = [SiLU](GroupNorm(x + add[:, :, None, None])): nn.GroupNorm (+ the time-embedding add in front and the
    nn.SiLU behind it, residual.py:226-247) in one kernel each way."""

    @staticmethod
    def forward(ctx: Any, x: Tensor, weight: Tensor, bias: Tensor, groups: int, eps: float, add: Optional[Tensor],
                silu: bool) -> Tensor:
        if x.dtype not in (bf16, f32):
            x = x.float()
        gamma, beta = weight.detach().contiguous(), bias.detach().contiguous()
        addc = None if add is None else add.detach().float().contiguous()
        ctx.weight, ctx.bias, ctx.groups, ctx.silu = weight, bias, groups, silu
        ctx.nhwc = None
        if is_nhwc(x) and x.dtype == bf16 and x.shape[1] % 8 == 0 and x.shape[1] <= 4096 and groups <= 64:
            # NHWC rows in, NHWC rows out (cfhip_groupnorm_nhwc_*): what sits between two implicit-GEMM convolutions
            b, c, h, w = x.shape
            rows = nhwc_rows(x)
            y_rows, mean, rstd = ops.groupnorm_nhwc_fwd(rows, b, gamma, beta, groups, eps, add=addc, silu=silu)
            ctx.save_for_backward(rows, gamma, beta, mean, rstd, addc)
            ctx.nhwc = (b, c, h, w)
            return rows_to_nhwc(y_rows, b, c, h, w)
        x = to_nchw(x)
        y, mean, rstd = ops.groupnorm_fwd(x, gamma, beta, groups, eps, add=addc, silu=silu)
        ctx.save_for_backward(x, gamma, beta, mean, rstd, addc)
        return y

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        x, gamma, beta, mean, rstd, addc = ctx.saved_tensors
        dy = as_bf16_act(dy)
        per_sample = gamma.dim() == 2  # one affine per sample: dgamma / dbeta stay [B, C] (ScaleShiftAffineFn reduces them)
        if ctx.nhwc is not None:
            b, c, h, w = ctx.nhwc
            dx_rows, dg, db, dadd = ops.groupnorm_nhwc_bwd(nhwc_rows(to_nhwc(dy)), x, b, gamma, beta, mean, rstd, ctx.groups, add=addc,
                                                           silu=ctx.silu)
            dx = rows_to_nhwc(dx_rows, b, c, h, w)
        else:
            dx, dg, db, dadd = ops.groupnorm_bwd(to_nchw(dy), x, gamma, beta, mean, rstd, ctx.groups, add=addc, silu=ctx.silu,
                                                 reduce=per_sample)
        gw = gb = None
        # both [B, C] partial sums reduced over B straight into `.grad` by ONE launch when the two parameters take the same kind of write
        if (not per_sample and ctx.weight.requires_grad and ctx.bias.requires_grad and _is_direct(ctx.weight) and _is_direct(ctx.bias)
                and dg.shape == db.shape and dg.dim() == 2
                and write_param_grad_pair(ctx.weight, ctx.bias,
                                          lambda ga, gb_, acc: ops.colreduce2_f32(dg, db, ga.view(-1), gb_.view(-1), acc))):
            return (dx if ctx.needs_input_grad[0] else None), None, None, None, None, dadd, None
        for prm, g, which in ((ctx.weight, dg, 0), (ctx.bias, db, 1)):
            if not prm.requires_grad:
                continue
            if per_sample:
                pass
            elif _is_direct(prm):  # the [B, C] partial sums are reduced over B straight into `.grad`
                write_param_grad(prm, lambda out, acc, g=g: ops.colreduce_f32(g, out=out.view(-1), accumulate=acc))
                continue
            else:
                g = ops.colreduce_f32(g)
            if _is_direct(prm):
                write_param_grad(prm, lambda out, acc, g=g: out.add_(g.view(out.shape)) if acc else out.copy_(g.view(out.shape)))
            elif which == 0:
                gw = g.view(prm.shape)
            else:
                gb = g.view(prm.shape)
        return (dx if ctx.needs_input_grad[0] else None), gw, gb, None, None, dadd, None


def group_norm(x: Tensor, weight: Tensor, bias: Tensor, groups: int, eps: float, add: Optional[Tensor] = None,
               silu: bool = False) -> Tensor:
    """`weight` / `bias`: [C], or [B, C] for one affine per sample (see `scale_shift_affine`)."""
    return _apply(GroupNormFn, x, weight, bias, groups, eps, add, silu)


class ScaleShiftAffineFn(Function):
    """The scale-shift norm `norm(net) * (1 + scale) + shift` (reference residual.py:236-239) as GroupNorm with one affine
    per sample: gamma_eff[b, c] = gamma[c] (1 + scale[b, c]), beta_eff[b, c] = beta[c] (1 + scale[b, c]) + shift[b, c]
    (f32 [B, C]: a few hundred values; the normalisation itself stays one kernel).  The parameter gradients follow the
    package's direct-write protocol."""

    @staticmethod
    def forward(ctx: Any, weight: Tensor, bias: Tensor, scale: Tensor, shift: Tensor):  # type: ignore
        g, b_ = weight.detach().float(), bias.detach().float()
        sc = scale.detach().float()
        one_plus = sc + 1.0
        ctx.save_for_backward(g, b_, one_plus)
        ctx.weight, ctx.bias = weight, bias
        return (g[None, :] * one_plus).contiguous(), (b_[None, :] * one_plus + shift.detach().float()).contiguous()

    @staticmethod
    def backward(ctx: Any, dge: Tensor, dbe: Tensor):  # type: ignore
        g, b_, one_plus = ctx.saved_tensors
        dge, dbe = dge.float(), dbe.float()
        dgamma, dbeta = (dge * one_plus).sum(0), (dbe * one_plus).sum(0)
        gw = gb = None
        for prm, gr, which in ((ctx.weight, dgamma, 0), (ctx.bias, dbeta, 1)):
            if not prm.requires_grad:
                continue
            if _is_direct(prm):
                write_param_grad(prm, lambda out, acc, gr=gr: out.add_(gr.view(out.shape)) if acc else out.copy_(gr.view(out.shape)))
            elif which == 0:
                gw = gr.view(prm.shape)
            else:
                gb = gr.view(prm.shape)
        dscale = dge * g[None, :] + dbe * b_[None, :]
        return gw, gb, dscale, dbe


def scale_shift_affine(weight: Tensor, bias: Tensor, scale: Tensor, shift: Tensor):
    return _apply(ScaleShiftAffineFn, weight, bias, scale, shift)


class SiLUF32Fn(Function):
    """nn.SiLU on the small fp32 time embedding [B, C] (residual.py:227)"""

    @staticmethod
    def forward(ctx: Any, x: Tensor) -> Tensor:
        x = x.float().contiguous()
        ctx.save_for_backward(x)
        return ops.silu_f32_fwd(x)

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        (x,) = ctx.saved_tensors
        return ops.silu_f32_bwd(dy.float().contiguous(), x)


def silu_f32(x: Tensor) -> Tensor:
    return _apply(SiLUF32Fn, x)


# ---- the residual blocks' time-embedding projections, every block of a UNet forward in one launch (round 6) ------------------------------
# Each ResidualBlock computes Linear_i(SiLU(time_net)) from the same [B, K] embedding: per block a SiLU, a cast and an M = B GEMM on the
# critical queue, and in backward a cast, a GEMM, a copy, a SiLU' and autograd's add into the shared gradient (~160 launches per 64^2 x 8
# UNet step for 0.05 ms of arithmetic).  `time_proj_all` computes all of them at the top of the forward (cfhip_time_proj_fwd / _bwd); a
# block picks its own output up with `time_pre_lookup`.  The weight / bias gradients keep their path (one dW + db launch per block, on the
# side stream or in a grouped launch).
TIME_PROJ_GROUPED = True  # (A/B: tools set the attribute)


class _TimePre(threading.local):
    def __init__(self) -> None:
        self.src: Optional[Tensor] = None
        self.outs: dict = {}


_TIME_PRE = _TimePre()


class TimeProjAllFn(Function):
    """(emb, weights..., biases...) -> (Linear_i(SiLU(emb)) ..., bf16 SiLU(emb)).  Its backward is the EMBEDDING's gradient only (two
    launches when every block's dY is there, i.e. at the end of the UNet's backward); the weights' gradients are computed by each block's
    `TimeProjTapFn` when ITS dY arrives — an arena range of the in-backward optimizer must not wait for the end of the pass."""
    tapeable = False

    @staticmethod
    def forward(ctx: Any, emb: Tensor, count: int, *wb: Any):  # wb: count weights, then count biases (None allowed)
        weights = [whole_param(w) for w in wb[:count]]
        biases = [whole_param(b) for b in wb[count:]]
        w16 = [shadow_bf16(w).view(w.shape[0], -1) for w in weights]
        bias_f = [None if b is None else b.detach().reshape(-1).contiguous() for b in biases]
        emb = emb.contiguous()
        t16 = torch.empty(emb.shape, dtype=bf16, device=emb.device)
        outs = ops.time_proj_fwd(emb, w16, bias_f, t16)
        ctx.save_for_backward(emb, *w16)
        ctx.count = count
        ctx.mark_non_differentiable(t16)
        return tuple(outs) + (t16,)

    @staticmethod
    def backward(ctx: Any, *dys: Any):  # type: ignore
        emb = ctx.saved_tensors[0]
        w16 = ctx.saved_tensors[1:]
        count = ctx.count
        if not ctx.needs_input_grad[0]:
            return (None,) * (2 + 2 * count)
        dys = [None if dy is None else dy.float().contiguous() for dy in dys[:count]]
        d_emb = ops.time_proj_bwd(emb, w16, dys, [None] * count)
        return (d_emb, None) + (None,) * (2 * count)


class TimeProjTapFn(Function):
    """identity on one block's precomputed projection; in backward the block's weight / bias gradient (dW = dY^T SiLU(emb), db = colsum dY:
    on the side stream when they land straight in `.grad`) and dY handed on to `TimeProjAllFn`"""
    tapeable = False

    @staticmethod
    def forward(ctx: Any, pre: Tensor, t16: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
        ctx.weight, ctx.bias = whole_param(weight), whole_param(bias)
        ctx.save_for_backward(t16)
        return pre.view_as(pre)

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        (t16,) = ctx.saved_tensors
        w, b = ctx.weight, ctx.bias
        gw = gb = None
        if w.requires_grad or (b is not None and b.requires_grad):
            dyf = dy.float().contiguous()
            if _all_direct(w, b):
                SideStream.run(lambda: _linear_param_grads(ops.to_bf16(dyf), t16, w, b, True, True), (dyf, t16))
            else:
                gw, gb = _linear_param_grads(ops.to_bf16(dyf), t16, w, b, _is_direct(w), _is_direct(b))
        return dy, None, gw, gb


def time_proj_all(time_net: Tensor, blocks: Sequence[Any]) -> bool:
    """Linear_i(SiLU(time_net)) of every block in `blocks` (modules with a plain `time_embedding` Linear) in one launch, kept for
    `time_pre_lookup` until `time_pre_clear`.  False (nothing done) when the grouped kernel does not take this input."""
    time_pre_clear()
    # (taped nodes — one autograd node per block, off by default — run each block's Functions on a tape of their own: per-block path)
    if (not TIME_PROJ_GROUPED or TAPED_NODES[0] or not blocks or getattr(_TAPE, "tape", None) is not None or not time_net.is_cuda or time_net.dtype != f32
            or time_net.dim() != 2 or time_net.shape[1] % 32 != 0 or time_net.shape[1] > 2048):
        return False
    pres: dict = {}
    for i in range(0, len(blocks), ops.TIME_PROJ_MAX):
        chunk = blocks[i:i + ops.TIME_PROJ_MAX]
        ws = [blk.time_embedding.weight for blk in chunk]
        bs = [blk.time_embedding.bias for blk in chunk]
        res = TimeProjAllFn.apply(time_net, len(chunk), *ws, *bs)
        for blk, out in zip(chunk, res[:-1]):
            pres[id(blk)] = (out, res[-1])
    _TIME_PRE.src = time_net
    _TIME_PRE.outs = pres
    return True


def time_pre_lookup(block: Any, time_net: Tensor) -> Optional[Tensor]:
    """`block.time_embedding(SiLU(time_net))` out of the grouped launch for exactly this `time_net` tensor, else None"""
    if _TIME_PRE.src is not time_net or getattr(_TAPE, "tape", None) is not None:
        return None
    hit = _TIME_PRE.outs.get(id(block))
    if hit is None:
        return None
    lin = block.time_embedding
    return TimeProjTapFn.apply(hit[0], hit[1], lin.weight, lin.bias)


def time_pre_clear() -> None:
    _TIME_PRE.src = None
    _TIME_PRE.outs = {}


class Upsample2Fn(Function):
    """F.interpolate(scale_factor=2, mode="nearest") (residual.py:147)"""

    @staticmethod
    def forward(ctx: Any, x: Tensor) -> Tensor:
        x = as_bf16_act(x)
        ctx.nhwc = is_nhwc(x) and x.shape[1] % 8 == 0
        if ctx.nhwc:
            b, c, h, w = x.shape
            return rows_to_nhwc(ops.upsample2_nhwc(nhwc_rows(x), b, h, w), b, c, 2 * h, 2 * w)
        return ops.upsample2_fwd(to_nchw(x))

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        dy = as_bf16_act(dy)
        if ctx.nhwc:
            b, c, h2, w2 = dy.shape
            return rows_to_nhwc(ops.upsample2_nhwc(nhwc_rows(to_nhwc(dy)), b, h2 // 2, w2 // 2, backward=True), b, c, h2 // 2, w2 // 2)
        return ops.upsample2_bwd(to_nchw(dy))


class AvgPool2Fn(Function):
    """nn.AvgPool2d(kernel_size=2, stride=2) (`ResDownsample(use_conv=False)`, residual.py:106)"""

    @staticmethod
    def forward(ctx: Any, x: Tensor) -> Tensor:
        x = as_bf16_act(x)
        ctx.nhwc = is_nhwc(x)  # (no NHWC pooling kernel: the zoo UNet down-samples with strided convolutions; one explicit hop each way)
        y = ops.avgpool2_fwd(to_nchw(x))
        return to_nhwc(y) if ctx.nhwc else y

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        dx = ops.avgpool2_bwd(to_nchw(as_bf16_act(dy)))
        return to_nhwc(dx) if ctx.nhwc else dx


class ReflectPad2dFn(Function):
    """nn.ReflectionPad2d (reference convs/basic.py:61-75: the `padding="reflection[N]"` form of Conv2d)"""

    @staticmethod
    def forward(ctx: Any, x: Tensor, pads: Tuple[int, int, int, int]) -> Tensor:
        if x.dtype not in (bf16, f32):
            x = x.float()
        ctx.pads = pads
        return ops.reflect_pad2d_fwd(to_nchw(x), pads)

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        return ops.reflect_pad2d_bwd(to_nchw(as_bf16_act(dy)), ctx.pads), None


def reflect_pad2d(x: Tensor, pads: Any) -> Tensor:
    """`pads`: one int or (left, right, top, bottom), as nn.ReflectionPad2d takes them"""
    if isinstance(pads, int):
        pads = (pads,) * 4
    return _apply(ReflectPad2dFn, x, tuple(int(v) for v in pads))


def upsample2(x: Tensor) -> Tensor:
    return _apply(Upsample2Fn, x)


def avg_pool2(x: Tensor) -> Tensor:
    return _apply(AvgPool2Fn, x)


# ---------------------------------------------------------------------------------------------
# SpatialTransformer glue (reference mixed_stacks/api.py:766-893): GEGLU, NCHW <-> token-major
# ---------------------------------------------------------------------------------------------


class GegluFn(Function):
    """value * gelu(gate) with [value | gate] = the two halves of the last dim (reference activations.py:150-158)"""

    @staticmethod
    def forward(ctx: Any, vg: Tensor) -> Tensor:
        vg = (vg if vg.dtype == bf16 else ops.to_bf16(vg.float().contiguous())).contiguous()
        ctx.save_for_backward(vg)
        return ops.geglu_fwd(vg)

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        (vg,) = ctx.saved_tensors
        dy = (dy if dy.dtype == bf16 else ops.to_bf16(dy.float().contiguous())).contiguous()
        return ops.geglu_bwd(dy, vg)


def geglu(vg: Tensor) -> Tensor:
    return _apply(GegluFn, vg)


class NchwToTokensFn(Function):
    """[B, C, H, W] -> token-major [B, H*W, C] (`permute(0, 2, 3, 1).reshape`), bf16"""

    @staticmethod
    def forward(ctx: Any, x: Tensor) -> Tensor:
        if x.dtype not in (bf16, f32):
            x = x.float()
        b, c, h, w = x.shape
        ctx.hw = (h, w)
        ctx.nhwc = is_nhwc(x) and x.dtype == bf16
        if ctx.nhwc:  # NHWC rows ARE the token-major matrix
            return nhwc_rows(x).view(b, h * w, c)
        return ops.transpose_batched(to_nchw(x).view(b, c, h * w))

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        b, t, c = dy.shape
        if ctx.nhwc:
            dy = dy if dy.dtype == bf16 else ops.to_bf16(dy.float().contiguous())
            return rows_to_nhwc(dy.contiguous().view(b * t, c), b, c, *ctx.hw)
        return ops.transpose_batched(dy.contiguous()).view(b, c, *ctx.hw)


class TokensToNchwFn(Function):
    """token-major [B, H*W, C] -> [B, C, H, W], bf16"""

    @staticmethod
    def forward(ctx: Any, x: Tensor, h: int, w: int) -> Tensor:
        if x.dtype not in (bf16, f32):
            x = x.float()
        b, t, c = x.shape
        if NHWC[0] and c % 8 == 0 and c > 1 and t > 1:  # the token-major matrix IS the image's NHWC rows: a view
            x = x if x.dtype == bf16 else ops.to_bf16(x.contiguous())
            return rows_to_nhwc(x.contiguous().view(b * t, c), b, c, h, w)
        return ops.transpose_batched(x.contiguous()).view(b, c, h, w)

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        b, c, h, w = dy.shape
        dy = as_bf16_act(dy)
        if is_nhwc(dy):
            return nhwc_rows(dy).view(b, h * w, c), None, None
        return ops.transpose_batched(dy.contiguous().view(b, c, h * w)), None, None


def nchw_to_tokens(x: Tensor) -> Tensor:
    return _apply(NchwToTokensFn, x)


def tokens_to_nchw(x: Tensor, h: int, w: int) -> Tensor:
    return _apply(TokensToNchwFn, x, h, w)


class ConcatChannelsFn(Function):
    """torch.cat([a, b], dim=1) for NCHW bf16 tensors (the UNet's skip connections, unet.py:311-316)"""

    @staticmethod
    def forward(ctx: Any, a: Tensor, b: Tensor) -> Tensor:
        ctx.nhwc = None
        if a.dim() == 4 and (is_nhwc(a) or is_nhwc(b)) and a.shape[1] > 1 and b.shape[1] > 1:
            # NHWC rows: every pixel row of the output is [a's channels | b's channels] — two strided row copies
            a, b = to_nhwc(a), to_nhwc(b)
            n, ca, h, w = a.shape
            cb = b.shape[1]
            out = torch.empty((n * h * w, ca + cb), dtype=bf16, device=a.device)
            ops.copy_strided2(nhwc_rows(a), out, ca, ca, ca + cb, nhwc_rows(b), out, cb, cb, ca + cb, n * h * w, dst_b_off=ca)
            ctx.nhwc = (n, ca, cb, h, w)
            return rows_to_nhwc(out, n, ca + cb, h, w)
        a = (a if a.dtype == bf16 else ops.to_bf16(a.float().contiguous())).contiguous()
        b = (b if b.dtype == bf16 else ops.to_bf16(b.float().contiguous())).contiguous()
        n, ca, cb = a.shape[0], a.shape[1], b.shape[1]
        inner = a.numel() // (n * ca)
        out = torch.empty((n, ca + cb, *a.shape[2:]), dtype=bf16, device=a.device)
        tot = (ca + cb) * inner
        ops.copy_strided2(a, out, ca * inner, ca * inner, tot, b, out, cb * inner, cb * inner, tot, n, dst_b_off=ca * inner)
        ctx.split = (ca, cb, inner)
        return out

    @staticmethod
    def backward(ctx: Any, dy: Tensor):  # type: ignore
        if ctx.nhwc is not None:
            n, ca, cb, h, w = ctx.nhwc
            rows = nhwc_rows(to_nhwc(as_bf16_act(dy)))
            da = torch.empty((n * h * w, ca), dtype=bf16, device=dy.device)
            db = torch.empty((n * h * w, cb), dtype=bf16, device=dy.device)
            ops.copy_strided2(rows, da, ca, ca + cb, ca, rows, db, cb, ca + cb, cb, n * h * w, src_b_off=ca)
            return rows_to_nhwc(da, n, ca, h, w), rows_to_nhwc(db, n, cb, h, w)
        ca, cb, inner = ctx.split
        dy = to_nchw(dy if dy.dtype == bf16 else ops.to_bf16(to_nchw(dy.float())))
        n = dy.shape[0]
        da = torch.empty((n, ca, *dy.shape[2:]), dtype=bf16, device=dy.device)
        db = torch.empty((n, cb, *dy.shape[2:]), dtype=bf16, device=dy.device)
        tot = (ca + cb) * inner
        ops.copy_strided2(dy, da, ca * inner, tot, ca * inner, dy, db, cb * inner, tot, cb * inner, n, src_b_off=ca * inner)
        return da, db


def concat_channels(a: Tensor, b: Tensor) -> Tensor:
    return _apply(ConcatChannelsFn, a, b)


# ---------------------------------------------------------------------------------------------
# gradient checkpointing (reference toolkit.py:2535-2647 `gradient_checkpoint`: `use_checkpoint=True` in the zoo
# `diffusion/ddpm` config wraps every ResidualBlockWithTimeEmbedding / SpatialTransformerBlock)
# ---------------------------------------------------------------------------------------------


class _CheckpointFn(Function):
    """Runs `fn` without recording in forward (nothing inside is saved for backward) and again, recording, inside
    backward; the gradients of that second run are what flows out.  The HIP Functions inside are re-entered under
    `torch.autograd.grad`: their parameter gradients are written straight into `.grad` (arena) during the inner
    backward and returned as None, so this node returns None for every parameter as well — nothing is counted twice.
    With 288 GB of HBM per GPU the default stays OFF; the switch is for the configurations that set it."""

    tapeable = False  # re-enters autograd: not inside a taped node

    @staticmethod
    def forward(ctx: Any, fn: Callable, n_inputs: int, *args: Any) -> Any:  # type: ignore
        ctx.fn = fn
        ctx.inputs = list(args[:n_inputs])
        ctx.params = list(args[n_inputs:])
        ctx.requires = [isinstance(x, Tensor) and x.requires_grad for x in ctx.inputs]
        ctx.nhwc = NHWC[0]  # the recomputation runs inside backward, after UNetDiffuser.forward has restored the flag: same hand-over
        with torch.no_grad():
            return fn(*ctx.inputs)

    @staticmethod
    def backward(ctx: Any, *grad_outputs: Any) -> Any:  # type: ignore
        for cb in backward_entered_callbacks:  # outer graph task: see the note at the list's definition
            cb()
        inputs = [x.detach().requires_grad_(r) if isinstance(x, Tensor) else x for x, r in zip(ctx.inputs, ctx.requires)]
        keep = NHWC[0]
        NHWC[0] = ctx.nhwc
        try:
            with torch.enable_grad():
                outputs = ctx.fn(*[x.view_as(x) if isinstance(x, Tensor) else x for x in inputs])
        finally:
            NHWC[0] = keep
        if isinstance(outputs, Tensor):
            outputs = (outputs,)
        wrt = [x for x, r in zip(inputs, ctx.requires) if r]
        wrt_params = [p for p in ctx.params if p.requires_grad]
        grads = torch.autograd.grad(outputs, wrt + wrt_params, grad_outputs, allow_unused=True)
        it = iter(grads[:len(wrt)])
        in_grads = [next(it) if r else None for r in ctx.requires]
        it = iter(grads[len(wrt):])
        p_grads = [next(it) if p.requires_grad else None for p in ctx.params]
        ctx.inputs = ctx.params = None
        return (None, None) + tuple(in_grads) + tuple(p_grads)


def gradient_checkpoint(fn: Callable, inputs: Any, params: Any, enabled: bool) -> Any:
    """Same call signature and semantics as the reference's `gradient_checkpoint(func, inputs, params, enabled)`."""
    if not enabled or not torch.is_grad_enabled():
        return fn(*inputs)
    inputs = tuple(inputs)
    return _apply(_CheckpointFn, fn, len(inputs), *(inputs + tuple(params)))


# ---------------------------------------------------------------------------------------------
# A12: dropout / DropPath as autograd Functions.  Forward draws (seed, offset) from ops.PhiloxState and remembers the
# pair; backward runs the SAME kernel on dy with the same pair (the mask is regenerated, never stored).
# ---------------------------------------------------------------------------------------------


class DropoutFn(Function):
    @staticmethod
    def forward(ctx: Any, x: Tensor, p: float, mask: Optional[Tensor]) -> Tensor:  # type: ignore
        ctx.p, ctx.mask = float(p), mask
        ctx.seed, ctx.offset = (0, 0) if mask is not None else ops.PhiloxState.take((x.numel() + 3) // 4)
        # the mask is indexed in the LOGICAL (NCHW) element order — what an injected mask and torch's own use: NHWC rows hop over
        ctx.nhwc = x.dim() == 4 and is_nhwc(x)
        y, _ = ops.dropout(to_nchw(x) if ctx.nhwc else x, ctx.p, seed=ctx.seed, offset=ctx.offset, mask=mask)
        return to_nhwc(y) if ctx.nhwc else y

    @staticmethod
    def backward(ctx: Any, dy: Tensor) -> Any:  # type: ignore
        dx, _ = ops.dropout(to_nchw(dy) if dy.dim() == 4 else dy.contiguous(), ctx.p, seed=ctx.seed, offset=ctx.offset, mask=ctx.mask)
        return (to_nhwc(dx) if ctx.nhwc else dx), None, None


def dropout(x: Tensor, p: float, training: bool, mask: Optional[Tensor] = None) -> Tensor:
    """nn.Dropout semantics (identity unless training and 0 < p < 1); `mask` (uint8, 1 = keep) injects the mask."""
    if not training or not 0.0 < p < 1.0:
        return x
    return _apply(DropoutFn, x, p, mask)


class DropPathFn(Function):
    @staticmethod
    def forward(ctx: Any, x: Tensor, keep_prob: float, mask: Tensor) -> Tensor:  # type: ignore
        ctx.keep_prob = float(keep_prob)
        ctx.save_for_backward(mask)
        return ops.drop_path(x, mask, ctx.keep_prob)

    @staticmethod
    def backward(ctx: Any, dy: Tensor) -> Any:  # type: ignore
        (mask,) = ctx.saved_tensors
        return ops.drop_path(dy.contiguous(), mask, ctx.keep_prob), None, None


def drop_path(x: Tensor, rate: float, training: bool, mask: Optional[Tensor] = None) -> Tensor:
    """reference customs.py:434-443: per-sample stochastic depth; `mask` (f32 [B] of 0 / 1) injects the sample mask."""
    if not training or not 0.0 < rate < 1.0:
        return x
    keep_prob = 1.0 - rate
    if mask is None:
        b = x.shape[0]
        seed, offset = ops.PhiloxState.take((b + 3) // 4)
        mask = ops.drop_path_mask(b, keep_prob, x.device, seed=seed, offset=offset)
    return _apply(DropPathFn, x, keep_prob, mask)


# ---------------------------------------------------------------------------------------------
# A16: tabular encoder
# ---------------------------------------------------------------------------------------------


class MLEncodeFn(Function):
    @staticmethod
    def forward(ctx: Any, x: Tensor, plan: Tensor, tables_ptr: Optional[Tensor], out_dim: int, *tables: Tensor) -> Tensor:  # type: ignore
        ctx.save_for_backward(x, plan)
        ctx.tables = tables
        return ops.ml_encode_fwd(x, plan, tables_ptr, out_dim)

    @staticmethod
    def backward(ctx: Any, dout: Tensor) -> Any:  # type: ignore
        x, plan = ctx.saved_tensors
        grads = [torch.zeros_like(t) if t.requires_grad else None for t in ctx.tables]
        ptrs = None
        if grads:
            ptrs = torch.tensor([0 if g is None else g.data_ptr() for g in grads], dtype=torch.int64).to(x.device)
        dx = ops.ml_encode_bwd(dout.float(), x, plan, ptrs, ctx.needs_input_grad[0])
        return (dx, None, None, None) + tuple(grads)
