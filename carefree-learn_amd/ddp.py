"""Data-parallel gradient exchange: bucketed all-reduce on RCCL over xGMI, overlapped with backward.

What it replaces: the torch-DDP reducer that `accelerator.prepare` installs in the reference
(trainer.py:268-272) — and which the reference then bypasses (`model_for_training` is
never called, so its autograd hooks never arm).  The reference's intent is DDP-mean semantics
(g <- sum_r g_r / W once per optimizer step); this module implements that intent by hooking
PARAMETERS (gradient-ready notifications from the HIP backward kernels, plus autograd
post-accumulate hooks for everything else), never a wrapper in the forward path.

MI355X specifics: one process per GPU; the buckets are zero-copy slices of the gradient arena
(`optim.ParamArena.flat_g`), built in reverse registration order (≈ backward completion order) and
sized >= 32 MB so that an xGMI link (≈153 GB/s, point-to-point, 7 per GPU) is bandwidth- rather
than latency-bound; each bucket's all-reduce is issued on a side HIP stream as soon as its last
gradient lands and the optimizer waits on the events.  The 1/W averaging costs nothing: it is the
`grad_scale` of the fused Adam kernel.  `torch.distributed` (backend "nccl" == RCCL on ROCm; "gloo"
in the CPU tests) is used for rendezvous and the collective launch.
"""
import contextlib
from typing import Any, Iterator, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor

from . import functional as HF
from .optim import FusedAdam, ParamArena


class _Bucket:
    __slots__ = ("start", "end", "param_ids", "pending", "work", "launched")

    def __init__(self, start: int, end: int, param_ids: List[int]):
        self.start, self.end, self.param_ids = start, end, param_ids
        self.pending = len(param_ids)
        self.work: Any = None
        self.launched = False


def get_ddp_info() -> Optional[dict]:
    """RANK / WORLD_SIZE / LOCAL_RANK from the launcher's environment (reference toolkit.py:1902-1921)."""
    import os

    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and "LOCAL_RANK" in os.environ:
        return dict(rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]),
                    local_rank=int(os.environ["LOCAL_RANK"]))
    return None


class Communicator:
    """RCCL communicator owned by this package through the C-ABI (`cfhip_comm_*`, include/cfhip.h): collectives are
    launched on the stream the CALLER names — the comm stream that `functional.distinct_stream` has checked for a
    hardware queue of its own — instead of the internal stream of a torch ProcessGroup.  The 128-byte RCCL unique id
    travels through the already-initialised torch.distributed group (rendezvous only: one broadcast at start-up)."""

    def __init__(self, group: Any = None):
        import ctypes

        from . import _lib

        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("Communicator needs torch.distributed for the rendezvous of the RCCL unique id")
        self._lib = _lib
        lib = _lib.load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        uid = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (ctypes.c_char * 128)()
            _lib.check(lib.cfhip_comm_unique_id(buf), "comm_unique_id")
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if self.world > 1:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
            uid = uid.to(dev)
            dist.broadcast(uid, src=0, group=group)
            uid = uid.cpu()
        handle = ctypes.c_void_p()
        raw = (ctypes.c_char * 128).from_buffer_copy(bytes(uid.tolist()))
        _lib.check(lib.cfhip_comm_init(self.rank, self.world, raw, ctypes.byref(handle)), "comm_init")
        self.handle = handle

    def count(self) -> tuple:
        """(ranks, this rank) as RCCL reports them for the communicator (ncclCommCount / ncclCommUserRank)"""
        import ctypes

        world, rank = ctypes.c_int(-1), ctypes.c_int(-1)
        self._lib.check(self._lib.load().cfhip_comm_count(self.handle, ctypes.byref(world), ctypes.byref(rank)), "comm_count")
        return world.value, rank.value

    @staticmethod
    def _dt(t: Tensor) -> int:
        if t.dtype == torch.float32:
            return 0
        if t.dtype == torch.bfloat16:
            return 1
        raise TypeError(f"cfhip comm: f32 or bf16 tensors only, got {t.dtype}")

    def all_reduce_(self, t: Tensor, stream: "torch.cuda.Stream") -> None:
        self._lib.check(self._lib.load().cfhip_comm_allreduce(self.handle, t.data_ptr(), t.numel(), self._dt(t),
                                                              stream.cuda_stream), "comm_allreduce")

    def broadcast_(self, t: Tensor, root: int, stream: "torch.cuda.Stream") -> None:
        self._lib.check(self._lib.load().cfhip_comm_broadcast(self.handle, t.data_ptr(), t.numel(), self._dt(t), root,
                                                              stream.cuda_stream), "comm_broadcast")

    def all_gather(self, send: Tensor, recv: Tensor, stream: "torch.cuda.Stream") -> None:
        self._lib.check(self._lib.load().cfhip_comm_allgather(self.handle, send.data_ptr(), recv.data_ptr(), send.numel(),
                                                              self._dt(send), stream.cuda_stream), "comm_allgather")

    def reduce_scatter(self, send: Tensor, recv: Tensor, stream: "torch.cuda.Stream") -> None:
        self._lib.check(self._lib.load().cfhip_comm_reduce_scatter(self.handle, send.data_ptr(), recv.data_ptr(),
                                                                   recv.numel(), self._dt(recv), stream.cuda_stream),
                        "comm_reduce_scatter")

    def close(self) -> None:
        if self.handle is not None:
            self._lib.check(self._lib.load().cfhip_comm_destroy(self.handle), "comm_destroy")
            self.handle = None


class _StreamWork:
    """`work.wait()` of a collective launched through the C-ABI: the compute stream waits for an event on the comm stream"""

    def __init__(self, stream: "torch.cuda.Stream"):
        self.event = torch.cuda.Event()
        self.event.record(stream)

    def wait(self) -> None:
        torch.cuda.current_stream().wait_event(self.event)


class BucketedAllReduce:
    """`average`: True = `finish()` leaves the rank-AVERAGED gradients in the arena (one extra pass over it; what a
    trainer that clips / inspects gradients between backward and `optimizer.step()` needs); False = the arena holds the
    SUM and the fused Adam kernel applies 1 / W as its `grad_scale` (engine.TrainStep).  Default: False when a fused
    optimizer is given, True otherwise.
    `finish_after_backward`: queue `finish()` as an autograd end-of-backward callback (first gradient notification of
    a backward pass), so that whatever runs between `backward()` and `optimizer.step()` sees reduced gradients; the
    explicit `finish()` (optimizer pre-step hook, `TrainStep`) then finds nothing left to do.
    `sync_fn`: evaluated once per backward pass; False = this pass only accumulates locally (gradient accumulation,
    reference schema.py:1277-1282) — nothing is launched and `finish()` is a no-op."""

    def __init__(self, arena: ParamArena, *, process_group: Any = None, bucket_bytes: int = 64 << 20,
                 overlap: bool = True, optimizer: Optional[FusedAdam] = None, average: Optional[bool] = None,
                 finish_after_backward: bool = False, sync_fn: Any = None, wire_bf16: bool = False,
                 comm: Optional["Communicator"] = None, tail_bytes: int = 4 << 20, step_in_backward: bool = False):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("BucketedAllReduce needs an initialised torch.distributed process group")
        self.arena = arena
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.overlap = overlap
        self.sync_enabled = True
        self.sync_fn = sync_fn
        self.optimizer = optimizer
        self.average = (optimizer is None) if average is None else bool(average)
        self.finish_after_backward = finish_after_backward
        self.wire_bf16 = wire_bf16
        # the fused Adam(W) update of a bucket's arena range right behind its all-reduce, on the comm stream (the arena holds
        # the rank SUM and the kernel applies 1 / W): see optim.StepInBackward for why no running kernel sees a half-updated
        # weight.  Not with bf16 on the wire (widened in finish()), not with an averaged arena (a trainer that clips).
        self._want_step_in_backward = bool(step_in_backward) and optimizer is not None and arena.flat_g.is_cuda
        self.step_in_backward = self._want_step_in_backward and not wire_bf16
        if self._want_step_in_backward:
            arena.double_buffer_shadows()
        self.comm = comm  # None: torch.distributed launches the collectives; a Communicator: the cfhip_comm_* C-ABI
        self.is_cuda = arena.flat_g.is_cuda
        self.comm_stream = None
        if self.is_cuda:
            # helper streams are checked for their own hardware queue (functional.distinct_stream): created here,
            # BEFORE the first collective makes the process group create its internal stream
            HF.SideStream.ensure()
            self.comm_stream = HF.distinct_stream(HF.SideStream.streams)
        # buckets: walk the parameters from LAST to FIRST (backward produces them in that order).  The bucket that
        # closes LAST (the first-registered parameters: stem / positional encoding, whose gradients appear when backward
        # ends) is the one exchange nothing can hide: it is kept small (`tail_bytes`) — the parameters before the first
        # cut-off go into a bucket of their own instead of riding on up to `bucket_bytes` of earlier-finished gradients.
        self.tail_bytes = tail_bytes
        self._index = {id(p): i for i, p in enumerate(arena.params)}
        self._pass_open = False      # a backward pass has notified since the last finish()
        self.rebucket(bucket_bytes)
        self._pass_sync = True       # ... and it is a synchronising pass (sync_fn at its first notification)
        self._wire: Optional[Tensor] = None  # bf16 staging of the gradient arena (wire_bf16)
        self.exposed_events: List[Any] = []  # (start, end) event pairs around finish()'s waits, when timing is on
        self.time_exposed = False
        HF.grad_ready_callbacks.append(self._on_direct)
        HF.backward_entered_callbacks.append(self._on_backward_entered)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_ready) for p in arena.params]
        if optimizer is not None:
            optimizer.grad_scale = 1.0 if self.average else 1.0 / self.world_size

    def rebucket(self, bucket_bytes: int, tail_bytes: Optional[int] = None) -> int:
        """(Re)cut the gradient arena into buckets — the same communicator, streams and hooks (`bench.py` sweeps the bucket size
        of a live job with it: a new communicator per candidate would also re-read NCCL_* settings).  Only BETWEEN passes:
        every rank must call it with the same sizes at the same point of its step sequence.  Returns the bucket count."""
        if self._pass_open or any(b.launched for b in getattr(self, "buckets", [])):
            raise RuntimeError("BucketedAllReduce.rebucket: a backward pass is open (call it after finish() / the optimizer step)")
        arena = self.arena
        if tail_bytes is None:
            tail_bytes = self.tail_bytes
        self.bucket_bytes, self.tail_bytes = int(bucket_bytes), int(tail_bytes)
        self.buckets = []
        n = len(arena.params)
        n_tail, acc = 0, 0
        if tail_bytes > 0 and arena.total * 4 > bucket_bytes:
            while n_tail < n - 1 and acc < tail_bytes:
                acc += arena.params[n_tail].numel() * 4
                n_tail += 1
        ids: List[int] = []
        end = arena.total
        acc = 0
        for i in range(n - 1, -1, -1):
            ids.append(i)
            acc += arena.params[i].numel() * 4
            if acc >= bucket_bytes or i == 0 or i == n_tail:
                start = arena.offsets[i]
                self.buckets.append(_Bucket(start, end, ids))
                end, ids, acc = start, [], 0
        self.bucket_of = [0] * n
        for bi, b in enumerate(self.buckets):
            for i in b.param_ids:
                self.bucket_of[i] = bi
        self._ready = [False] * n
        self._direct = [False] * n  # notified by a HIP backward (functional.grad_ready_callbacks)
        return len(self.buckets)

    def set_wire_bf16(self, on: bool) -> None:
        """Switch the wire dtype of a live reducer (between passes; every rank alike).  bf16 on the wire keeps the optimizer at the
        end of the step (the widened sum exists only after finish())."""
        if self._pass_open:
            raise RuntimeError("BucketedAllReduce.set_wire_bf16: a backward pass is open")
        self.wire_bf16 = bool(on)
        self.step_in_backward = self._want_step_in_backward and not self.wire_bf16

    # -- life cycle ---------------------------------------------------------------------------
    def close(self) -> None:
        if self._on_direct in HF.grad_ready_callbacks:
            HF.grad_ready_callbacks.remove(self._on_direct)
        if self._on_backward_entered in HF.backward_entered_callbacks:
            HF.backward_entered_callbacks.remove(self._on_backward_entered)
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _broadcast(self, t: Tensor, src: int) -> None:
        if self.comm is not None and t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous():
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            self.comm.broadcast_(t, src, self.comm_stream)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            dist.broadcast(t, src=src, group=self.group)

    def broadcast_parameters(self, src: int = 0) -> None:
        """What the DDP constructor used to do: every rank starts from rank `src`."""
        self._broadcast(self.arena.flat_p, src)
        self.arena.refresh_shadow()

    def broadcast_buffers(self, modules: Any, src: int = 0) -> int:
        """The other half of the DDP constructor's `_sync_module_states` (torch DDP broadcasts parameters AND buffers at
        wrap time; behind trainer.py:268-272): BatchNorm running statistics, `num_batches_tracked`, attention masks ...
        of every module in `modules` follow rank `src`.  f32 buffers travel as ONE flat tensor.  Returns the number of
        buffers sent.
        Divergence from torch DDP: DDP's default `broadcast_buffers=True` repeats this at EVERY forward, so
        BatchNorm running statistics are rank 0's everywhere; here they are synchronised when the callback is installed and
        whenever the caller asks again (`RcclDDPCallback.sync_buffers()`, e.g. before evaluation or a checkpoint) — in between
        each rank keeps the statistics of its own shard, which does not enter the training arithmetic (train-mode BatchNorm
        normalises with batch statistics).  Buffers are expected on one device (the rank's GPU)."""
        seen, f32s, others = set(), [], []
        for m in modules:
            for b in m.buffers():
                if id(b) in seen or b.numel() == 0:
                    continue
                seen.add(id(b))
                (f32s if b.dtype == torch.float32 else others).append(b)
        devs = {b.device for b in f32s}
        if len(devs) > 1:
            raise ValueError(f"broadcast_buffers: buffers on several devices ({sorted(str(d) for d in devs)})")
        if f32s:
            flat = torch.cat([b.detach().reshape(-1) for b in f32s])
            self._broadcast(flat, src)
            off = 0
            with torch.no_grad():
                for b in f32s:
                    b.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()
        for b in others:
            t = b.detach().contiguous()
            t = t.to(torch.uint8) if t.dtype == torch.bool else t
            dist.broadcast(t, src=src, group=self.group)
            with torch.no_grad():
                b.copy_(t.to(b.dtype).view_as(b))
        return len(f32s) + len(others)

    @contextlib.contextmanager
    def no_sync(self) -> Iterator[None]:
        """Gradient accumulation: steps inside do not reduce (reference schema.py:1277-1282 — only
        the update step synchronises)."""
        prev, self.sync_enabled = self.sync_enabled, False
        try:
            yield
        finally:
            self.sync_enabled = prev

    # -- backward-time notifications ----------------------------------------------------------
    def _on_direct(self, p: Tensor) -> None:
        self._on_ready(p, True)

    def _on_backward_entered(self) -> None:
        """A checkpointed block is about to re-enter autograd (functional.backward_entered_callbacks): open the pass
        from HERE, the outer graph task, so that the end-of-backward callback belongs to the whole backward pass."""
        if not self._pass_open:
            self._open_pass()

    def _open_pass(self) -> None:
        """First gradient notification of a backward pass: decide whether this pass synchronises."""
        self._pass_open = True
        self._pass_sync = self.sync_fn is None or bool(self.sync_fn())  # `sync_enabled` (no_sync()) is read live
        if self.finish_after_backward:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
            except RuntimeError:  # not inside a backward pass: the explicit finish() does the work
                pass

    def _end_of_backward(self) -> None:
        for flush in HF.deferred_grad_flushes:  # weight gradients still queued for a grouped launch (fused.py)
            flush()
        if self._pass_open and self._pass_sync and self.sync_enabled:
            self.finish()
        self._pass_open = False

    def _on_ready(self, p: Tensor, direct: bool = False) -> None:
        i = self._index.get(id(p))
        if i is None:
            return
        if not self._pass_open:
            self._open_pass()
        if not direct and not self._direct[i]:
            # autograd accumulated this gradient: after the trainer's `optimizer.zero_grad()` (set_to_none) it sits
            # in a fresh tensor, not in the arena slot the bucket reduces
            self.arena.adopt_grad(p)
        if not (self._pass_sync and self.sync_enabled) or not self.overlap:
            return
        b = self.buckets[self.bucket_of[i]]
        if direct:
            self._direct[i] = True
        elif self._direct[i]:
            # autograd also runs the post-accumulate hook of a parameter whose gradient the HIP
            # backward wrote itself (the Function returned None for it): that is the echo of the
            # direct notification, not a second write
            return
        if self._ready[i]:
            if b.launched:
                raise RuntimeError(
                    f"parameter #{i} {tuple(p.shape)}: gradient written again after its bucket was reduced "
                    "(shared weights): construct BucketedAllReduce(overlap=False) for such models"
                )
            return
        self._ready[i] = True
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket) -> None:
        view = self.arena.flat_g[b.start:b.end]
        if self.wire_bf16:
            # bf16 on the wire (half the xGMI bytes): round the bucket into a staging arena, reduce that, widen back in
            # finish().  The sum of W bf16 values is exact in the f32 accumulate RCCL performs per element pair only up
            # to bf16 rounding of the partial sums — a numerics trade the caller opts into (default off).
            if self._wire is None:
                self._wire = torch.empty(self.arena.total, dtype=torch.bfloat16, device=view.device)
            wire = self._wire[b.start:b.end]
        if self.is_cuda:
            # The bucket's gradients were written by kernels on the current stream AND on the side
            # streams of functional.SideStream (dW GEMMs on one lane, column sums / LayerNorm parameter
            # gradients on another): the collective must be ordered after all of them.
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for side in HF.SideStream.streams:
                if side is not None:
                    self.comm_stream.wait_stream(side)
            with torch.cuda.stream(self.comm_stream):
                buf = view
                if self.wire_bf16:
                    wire.copy_(view)
                    buf = wire
                if self.comm is not None:
                    self.comm.all_reduce_(buf, self.comm_stream)
                    b.work = _StreamWork(self.comm_stream)
                else:
                    b.work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.step_in_backward and not self.average and self._pass_open and getattr(self.optimizer, "step_armed", False):
                    if self.comm is None:
                        b.work.wait()  # the process group reduces on a stream of its own: the comm stream follows it
                    self.optimizer.launch_range(b.start, b.end, self.comm_stream)
        else:
            if self.wire_bf16:
                wire.copy_(view)
                b.work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                b.work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        b.launched = True

    # -- before the optimizer step ----------------------------------------------------------------
    def finish(self) -> None:
        """Issue whatever has not been issued (unused parameters, overlap=False), then make the compute stream wait
        for every bucket.  Afterwards the arena holds the rank AVERAGE (`average=True`) or the rank SUM with 1 / W left
        to the optimizer's grad_scale.  Idempotent within one backward pass: the second call (e.g. the optimizer
        pre-step hook after the end-of-backward callback) returns at once; a pass that does not synchronise
        (`no_sync()`, `sync_fn` False) is left untouched."""
        if not self.sync_enabled or (self._pass_open and not self._pass_sync):
            self._pass_open = False
            return
        any_launched = any(b.launched for b in self.buckets)
        if not self._pass_open and not any_launched and self.overlap:
            return  # nothing was produced since the last finish()
        self._pass_open = False
        self.arena.finalize_grads()
        for b in self.buckets:  # fixed order on every rank
            if not b.launched:
                self._launch(b)
        ev0 = ev1 = None
        if self.time_exposed and self.is_cuda:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        for b in self.buckets:
            b.work.wait()
            if self.wire_bf16:
                self.arena.flat_g[b.start:b.end].copy_(self._wire[b.start:b.end])
            b.work, b.launched, b.pending = None, False, len(b.param_ids)
        if ev0 is not None:
            ev1.record()
            self.exposed_events.append((ev0, ev1))
        self._ready = [False] * len(self._ready)
        self._direct = [False] * len(self._direct)
        if self.average and self.world_size > 1:
            self.arena.flat_g.mul_(1.0 / self.world_size)

    def exposed_ms(self) -> List[float]:
        """Milliseconds the compute stream spent waiting for the exchange in each `finish()` since the last call (the
        'all-reduce ms (exposed)' column of BASELINE.md §4); needs `time_exposed = True`.  Synchronises."""
        out = []
        for e0, e1 in self.exposed_events:
            e1.synchronize()
            out.append(e0.elapsed_time(e1))
        self.exposed_events = []
        return out


class RcclDDPCallback:
    """Trainer-side seam (no trainer edits): duck-typed `TrainerCallback` whose `before_loop(trainer)`
    (reference trainer.py:312-313, schema.py:1755) re-homes the model's parameters into an arena,
    installs the bucketed all-reduce and broadcasts rank 0's weights.

    Ordering inside the reference's update (schema.py:977-986): `accelerator.backward(loss)` ->
    `trainer.clip_norm_step()` -> `optimizer.step()`.  Gradient clipping reads AND rescales the gradients, so the
    exchange must be complete — and averaged — when backward returns: `finish()` is queued as an end-of-backward
    callback and leaves the rank average in the arena; the optimizer pre-step hook stays as an idempotent fallback.
    Gradient accumulation (schema.py:1277-1282: update iff `state.step % grad_accumulate == 0`): passes that do not
    update only accumulate locally; the update pass reduces the accumulated sum once."""

    def __init__(self, bucket_bytes: int = 64 << 20, comm: str = "auto"):
        self.bucket_bytes = bucket_bytes
        self.comm_mode = comm  # "auto": the cfhip_comm_* C-ABI communicator under the nccl backend, else torch.distributed
        self.reducer: Optional[BucketedAllReduce] = None

    def sync_buffers(self, src: int = 0) -> int:
        """Every rank's module buffers (BatchNorm running statistics ...) <- rank `src`, again: call before evaluation or a
        checkpoint when the ranks must agree on them (torch DDP does this at every forward; see `broadcast_buffers`)."""
        if self.reducer is None:
            return 0
        return self.reducer.broadcast_buffers(getattr(self, "_buffer_modules", []), src)

    @staticmethod
    def _modules(trainer: Any) -> List[Any]:
        """Everything `accelerator.prepare(*model.all_modules, ...)` wrapped (trainer.py:268-272): the model's modules
        and, when it carries parameters or buffers of its own, the loss module (schema.py:1088-1091)."""
        model = trainer.model
        mods = [m for m in (getattr(model, "all_modules", None) or []) if isinstance(m, torch.nn.Module)]
        if not mods:
            mods = [model.m]
        loss = getattr(model, "loss", None)
        if isinstance(loss, torch.nn.Module) and all(loss is not m for m in mods):
            mods.append(loss)
        return mods

    def _communicator(self) -> Optional["Communicator"]:
        """The C-ABI communicator (collectives on this package's own queue-checked stream) when the process group is
        RCCL; every rank must agree, so a rank that cannot create it makes all of them fall back to torch.distributed."""
        if self.comm_mode == "torch" or not torch.cuda.is_available() or dist.get_backend() != "nccl":
            return None
        comm, ok = None, 1
        try:
            comm = Communicator()
        except Exception:  # missing RCCL symbol, communicator refused: the ProcessGroup launches the collectives
            ok = 0
        flag = torch.tensor([ok], device=torch.device("cuda", torch.cuda.current_device()), dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if comm is not None:
                comm.close()
            return None
        return comm

    @staticmethod
    def _sync_fn(trainer: Any) -> Any:
        def is_update_pass() -> bool:
            state = getattr(trainer, "state", None)
            config = getattr(trainer, "config", None)
            if state is None:
                return True
            steps = getattr(getattr(trainer, "model", None), "train_steps", None) or [None]
            for ts in steps:
                ga = getattr(ts, "grad_accumulate", None) or getattr(config, "grad_accumulate", 1) or 1
                if state.step % ga == 0:
                    return True
            return False

        return is_update_pass

    def before_loop(self, trainer: Any) -> None:
        if get_ddp_info() is None or not dist.is_initialized():
            return
        modules = self._modules(trainer)
        params, seen = [], set()
        for m in modules:
            for p in m.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    params.append(p)
        inners = [getattr(opt, "optimizer", opt) for opt in trainer.optimizers.values()]  # accelerate wraps the torch one
        # an optimizer that already owns the arena (optim.FusedAdamOptimizer): reduce ITS gradient buffer;
        # otherwise re-home the parameters here
        arena, fused = None, None
        for inner in inners:
            a = getattr(inner, "arena", None)
            if a is not None and {id(p) for p in a.params} == {id(p) for p in params}:
                arena, fused = a, getattr(inner, "fused", None)
                break
        if arena is None:
            owned = [p for p in params if getattr(p, "_cfhip_arena", None) is not None]
            if owned:
                raise RuntimeError(
                    f"RcclDDPCallback: {len(owned)} of the model's {len(params)} trainable parameters already live in "
                    "another ParamArena (several fused optimizers / scopes, or a frozen subset): re-homing them would "
                    "detach the fused Adam kernel from the buffers it updates.  Use one FusedAdam(W)Optimizer over "
                    "all trainable parameters, or a torch optimizer")
            arena = ParamArena(params, with_shadow=True)
        self.reducer = BucketedAllReduce(arena, bucket_bytes=self.bucket_bytes, optimizer=fused, average=True,
                                         finish_after_backward=True, sync_fn=self._sync_fn(trainer),
                                         comm=self._communicator())
        self.reducer.broadcast_parameters(0)
        self._buffer_modules = modules
        self.reducer.broadcast_buffers(modules, 0)
        for inner in inners:
            inner.register_step_pre_hook(lambda *_a, **_k: self.reducer.finish())
