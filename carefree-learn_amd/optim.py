"""Flat parameter / gradient arenas in HBM and the fused Adam(W) step over them.

Layout (sized for 288 GB HBM3E: full fp32 replica + fp32 grads + Adam moments + bf16 shadow =
18 B / parameter, 1.56 GB for ViT-B/16): every parameter of the model is a VIEW into one flat fp32
buffer, every `.grad` a view into a second one, every bf16 forward-pass copy a view into a third.
Consequences:
  * the optimiser step is ONE HBM-bound kernel over the arena (reads p, g, m, v; writes p, m, v and
    the bf16 shadow used by the next forward — no separate cast pass, no per-tensor launches);
  * DDP buckets are plain slices of the gradient arena (zero-copy all-reduce, `ddp.py`);
  * the backward kernels' fp32 outputs land directly in their final place.

The reference selects `torch.optim.Adam` / `AdamW` by name (optimizers.py:29-33) and steps it from
`get_update_fn` (schema.py:977-986); `FusedAdam` keeps that class's constructor / `step()` /
`zero_grad()` / `state_dict()` surface.
"""
import math
from typing import Any, Dict, Iterable, List, Optional, Tuple

import torch
from torch import Tensor

from . import ops

f32 = torch.float32
bf16 = torch.bfloat16
_HYPER_RING = 8  # the host may run this many optimizer steps ahead of the GPU
_ALIGN = 8  # elements: keeps every bf16 shadow view 16-byte aligned (MFMA GEMM operand rule)


class ParamArena:
    """Re-homes `params` (fp32, same device) into flat param / grad / bf16-shadow buffers."""

    def __init__(self, params: Iterable[Tensor], *, with_shadow: bool = True):
        self.params: List[Tensor] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("ParamArena: no trainable parameters")
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != f32 or p.device != dev:
                raise ValueError("ParamArena: parameters must be fp32 and on one device")
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = off
        self.flat_p = torch.zeros(off, dtype=f32, device=dev)
        self.flat_g = torch.zeros(off, dtype=f32, device=dev)
        self.flat_p16 = torch.zeros(off, dtype=bf16, device=dev) if with_shadow else None
        self.flat_p16_alt: Optional[Tensor] = None  # second shadow arena (update-in-backward: see double_buffer_shadows)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                p.grad = self.flat_g[o:o + n].view(p.shape)
                p._cfhip_arena = self   # functional.grad_buffer: a `.grad` that was set to None comes back as this view
                p._cfhip_fresh = False  # the arena starts zeroed: accumulate into it
                p.register_hook(self._fresh_guard(p))
                if self.flat_p16 is not None:
                    p._cfhip_shadow = self.flat_p16[o:o + n].view(p.shape)
                    p._cfhip_shadow_version = None  # filled by refresh_shadow()
        self.refresh_shadow()

    @staticmethod
    def _fresh_guard(p: Tensor):
        """Lazy zero-grad and gradients that arrive THROUGH AUTOGRAD (an op that was handed a slice / permutation of the
        parameter returns its gradient instead of writing `.grad`): autograd ADDS to `.grad`, so the slot's stale values
        must go first.  The tensor hook runs in front of the accumulation, only for such gradients (the direct-write
        kernels return None to autograd), and clears the 'fresh' mark so that `finalize_grads()` keeps the result."""

        def guard(grad: Tensor) -> None:
            if getattr(p, "_cfhip_fresh", False):
                if p.grad is not None:
                    p.grad.zero_()
                p._cfhip_fresh = False
            return None

        return guard

    def double_buffer_shadows(self) -> None:
        """A second bf16 shadow arena.  An optimizer that updates parameter ranges while backward is still running
        (`FusedAdam.enable_step_in_backward`) writes the NEW bf16 weights there: the dX GEMMs of this backward pass keep
        reading the arena the forward used, and `swap_shadows()` makes the new one current once the step is complete."""
        if self.flat_p16 is not None and self.flat_p16_alt is None:
            self.flat_p16_alt = self.flat_p16.clone()
            # both sets of per-parameter views exist up front: a swap is one attribute store per parameter
            self._shadow_views = [[p._cfhip_shadow for p in self.params],
                                  [self.flat_p16_alt[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.offsets)]]
            self._shadow_cur = 0

    def swap_shadows(self) -> None:
        """The shadow arena the optimizer has just filled becomes the one `functional.shadow_bf16` hands out."""
        if self.flat_p16_alt is None:
            return
        self.flat_p16, self.flat_p16_alt = self.flat_p16_alt, self.flat_p16
        self._shadow_cur ^= 1
        for p, v in zip(self.params, self._shadow_views[self._shadow_cur]):
            p._cfhip_shadow = v

    def refresh_shadow(self) -> None:
        """bf16 shadow <- fp32 masters (after construction / load_state_dict)."""
        if self.flat_p16 is None:
            return
        if self.flat_p.is_cuda:
            ops.to_bf16(self.flat_p, out=self.flat_p16)
        else:
            self.flat_p16.copy_(self.flat_p)
        for p in self.params:
            p._cfhip_shadow_version = p._version

    def zero_grad(self, lazy: bool = False) -> None:
        """Default: one memset over the gradient arena (autograd accumulates INTO `.grad`, so stale
        values must be gone).  `lazy=True` skips the memset and marks every gradient 'fresh' instead:
        the first HIP backward kernel that writes a parameter's gradient then OVERWRITES it and
        whatever nobody wrote is zeroed by `finalize_grads()`.  Gradients that autograd accumulates itself are covered by
        `_fresh_guard`."""
        if not lazy:
            self.flat_g.zero_()
        for p in self.params:
            if p.grad is None:  # somebody ran a set_to_none zero_grad: point it back into the arena
                self._rebind_grad(p)
            p._cfhip_fresh = lazy

    def _index(self, p: Tensor) -> int:
        idx = getattr(p, "_cfhip_arena_index", None)
        if idx is None:
            for i, q in enumerate(self.params):
                q._cfhip_arena_index = i
            idx = p._cfhip_arena_index
        return idx

    def grad_view(self, p: Tensor) -> Tensor:
        o = self.offsets[self._index(p)]
        return self.flat_g[o:o + p.numel()].view(p.shape)

    def _rebind_grad(self, p: Tensor) -> None:
        p.grad = self.grad_view(p)

    def adopt_grad(self, p: Tensor) -> None:
        """`p.grad` was produced outside the arena (the trainer called `optimizer.zero_grad()` with torch's default
        `set_to_none=True` and autograd allocated a fresh tensor): move it into the slot and point `.grad` back."""
        g = p.grad
        if g is None:
            return
        view = self.grad_view(p)
        if g.data_ptr() != view.data_ptr():
            view.copy_(g)
            p.grad = view

    def finalize_grads(self) -> None:
        """Zero the gradient slots no backward kernel touched this step (unused parameters)."""
        for p in self.params:
            if getattr(p, "_cfhip_fresh", False):
                p.grad.zero_()
                p._cfhip_fresh = False


class FusedAdam:
    """Adam / AdamW over a `ParamArena` in one kernel launch (K14)."""

    def __init__(self, params: Any, lr: float = 1.0e-3, betas: Any = (0.9, 0.999), eps: float = 1.0e-8,
                 weight_decay: float = 0.0, *, decoupled: bool = False, arena: Optional[ParamArena] = None):
        self.arena = arena if arena is not None else ParamArena(params)
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        self.param_groups = [dict(params=self.arena.params, **self.defaults)]
        self.decoupled = decoupled
        self.step_count = 0
        self.grad_scale = 1.0   # persistent factor on the gradients (1 / W when the arena holds the rank SUM)
        self._step_clip: Optional[Tensor] = None  # device scalar, ONE step only: the clipping coefficient
        self.lazy_zero = False  # see ParamArena.zero_grad
        a = self.arena
        self.exp_avg = torch.zeros_like(a.flat_p)
        self.exp_avg_sq = torch.zeros_like(a.flat_p)
        # The step-dependent scalars travel as a 32-byte record: pinned host slot -> async copy -> device record the
        # kernel reads.  The host runs ahead of the GPU (nothing in a training loop synchronises), so ONE host buffer
        # would be overwritten with step t+1's values before the queued copy of step t has executed (observed: the
        # loss trajectory of the UNet bench depended on launch timing from the 6th step on).  Hence a ring of slots,
        # each guarded by the event of the copy that last read it.
        self._hyper_ring = [torch.zeros(8, dtype=f32, pin_memory=a.flat_p.is_cuda) for _ in range(_HYPER_RING)]
        self._hyper_events: List[Any] = [None] * _HYPER_RING
        self._hyper_host = self._hyper_ring[0]
        self._hyper_dev = torch.zeros(8, dtype=f32, device=a.flat_p.device)
        self._done: List[Tuple[int, int]] = []  # arena ranges already updated in this step (update-in-backward)
        self._range_streams: List[Any] = []     # ... and the streams those updates were issued on
        self.in_backward: Optional["StepInBackward"] = None
        # True between prepare_step() and launch_step(): only then does `_hyper_dev` hold THIS step's record.  Whoever launches
        # range updates from inside backward (StepInBackward, ddp.BucketedAllReduce(step_in_backward=True)) checks it: with the
        # public `optimizer.step()` pattern (prepare + launch after backward) an early bucket would otherwise be updated with
        # the previous step's bias corrections — all zeros on step 1.
        self.step_armed = False

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.arena.zero_grad(lazy=self.lazy_zero)

    def enable_step_in_backward(self, range_bytes: int = 32 << 20) -> "StepInBackward":
        """Update each arena range as soon as every gradient in it is final, on a side stream, while backward is still
        running (the update is HBM-bound — 30 bytes per parameter — and backward's kernels are matrix-bound); what is left
        at `launch_step()` is the ranges that closed last.  See `StepInBackward` for what makes that safe."""
        if self.in_backward is None:
            self.in_backward = StepInBackward(self, range_bytes)
        return self.in_backward

    def launch_range(self, lo: int, hi: int, stream: Any = None) -> None:
        """The update of arena elements [lo, hi) (multiples of 8) on `stream` (default: current).  Element-wise: ranges in
        any order and number give bit-identical parameters to one launch over the arena."""
        a = self.arena
        if hi <= lo:
            return
        from . import _lib

        out16 = a.flat_p16_alt if a.flat_p16_alt is not None else a.flat_p16
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream.cuda_stream
        rc = _lib.load().cfhip_adam_step_dev(
            a.flat_p.data_ptr() + 4 * lo, a.flat_g.data_ptr() + 4 * lo, self.exp_avg.data_ptr() + 4 * lo,
            self.exp_avg_sq.data_ptr() + 4 * lo, None if out16 is None else out16.data_ptr() + 2 * lo, hi - lo,
            self._hyper_dev.data_ptr(), int(self.decoupled), st,
        )
        _lib.check(rc, "adam_step_dev")
        self._done.append((lo, hi))
        if stream is not None and all(stream != t for t in self._range_streams):
            self._range_streams.append(stream)

    def _fill_hyper(self) -> None:
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        t = self.step_count
        slot = t % _HYPER_RING
        ev = self._hyper_events[slot]
        if ev is not None:
            ev.synchronize()  # the copy that read this slot _HYPER_RING steps ago has executed
        h = self._hyper_host = self._hyper_ring[slot]
        h[0], h[1], h[2], h[3], h[4] = g["lr"], b1, b2, g["eps"], g["weight_decay"]
        h[5] = 1.0 - b1 ** t
        h[6] = 1.0 / math.sqrt(1.0 - b2 ** t)
        h[7] = self.grad_scale

    def prepare_step(self) -> None:
        """Host side of a step: advance t and upload the 32-byte hyper-parameter record.  Kept
        separate from `launch_step()` so the launch itself can live inside a captured hipGraph."""
        self.step_count += 1
        # ranges updated by a pass whose launch_step() never ran (an exception between the two, a skipped non-finite loss)
        # must not count as done in THIS step
        # — and their launches may still be queued on the comm / side stream, writing the masters, moments and shadows the next
        # full-arena launch on the current stream writes: order them first.  Such a pass leaves a PARTIAL update
        # behind (its launched ranges were stepped once with its gradients); nothing here can take that back.
        if self._range_streams and self._hyper_dev.is_cuda:
            cur = torch.cuda.current_stream()
            for st in self._range_streams:
                if st is not None and st != cur:
                    cur.wait_stream(st)
        self._done, self._range_streams = [], []
        self.step_armed = True  # this step's hyper-parameter record is on its way: range updates may be launched until launch_step()
        self._fill_hyper()
        self._hyper_dev.copy_(self._hyper_host, non_blocking=True)
        if self._hyper_dev.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._hyper_events[self.step_count % _HYPER_RING] = ev
        if self.in_backward is not None:
            self.in_backward.begin_step()

    def set_step_clip(self, coef: Tensor) -> None:
        """Gradient clipping for the NEXT `launch_step()` only: `coef` is a device scalar (<= 1) that multiplies the
        persistent `grad_scale` inside this step's hyper-parameter record ON THE DEVICE — no host read of the norm,
        nothing carried over to later steps (round 1 folded it into `grad_scale` itself: it compounded)."""
        if self._done:
            raise RuntimeError("FusedAdam: global-norm clipping needs every gradient before any update — do not enable "
                               "update-in-backward (enable_step_in_backward) together with clip_norm")
        self._step_clip = coef.reshape(1).to(f32)

    def launch_step(self) -> None:
        a = self.arena
        a.finalize_grads()
        if not a.flat_p.is_cuda:
            raise RuntimeError("FusedAdam: the arena must live on the HIP device (no CPU fallback)")
        from . import _lib

        if self._step_clip is not None:
            # after the upload of prepare_step() in stream order, before the kernel reads the record
            self._hyper_dev[7:8].copy_(self._step_clip * float(self.grad_scale))
            self._step_clip = None

        if self.in_backward is not None:
            self.in_backward.end_step()
        self.step_armed = False
        for st in self._range_streams:  # the caller's stream now follows every range update issued so far
            torch.cuda.current_stream().wait_stream(st)
        self._range_streams = []
        if not self._done:
            rc = _lib.load().cfhip_adam_step_dev(
                a.flat_p.data_ptr(), a.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                None if (a.flat_p16_alt if a.flat_p16_alt is not None else a.flat_p16) is None
                else (a.flat_p16_alt if a.flat_p16_alt is not None else a.flat_p16).data_ptr(),
                a.total, self._hyper_dev.data_ptr(), int(self.decoupled), torch.cuda.current_stream().cuda_stream,
            )
            _lib.check(rc, "adam_step_dev")
        else:  # what the in-backward updates left: the complement of the ranges already done
            done = sorted(self._done)
            self._done = []
            pos = 0
            for lo, hi in done:
                if lo > pos:
                    self.launch_range(pos, lo)
                pos = max(pos, hi)
            if pos < a.total:
                self.launch_range(pos, a.total)
            self._done = []
        a.swap_shadows()

    def step(self, closure: Any = None) -> None:
        self.prepare_step()
        self.launch_step()

    def snapshot(self) -> Dict[str, Any]:
        """Everything a step changes (masters, current bf16 shadows, moments, step counter): a hipGraph capture runs eager
        warm-up steps for the allocator and the lazy initialisations — `restore()` afterwards, so that the first replay is
        the FIRST update of its batch (three updates of the first batch, bias corrections advanced by 3)."""
        a = self.arena
        return dict(p=a.flat_p.clone(), p16=None if a.flat_p16 is None else a.flat_p16.clone(), m=self.exp_avg.clone(),
                    v=self.exp_avg_sq.clone(), step=self.step_count)

    def restore(self, snap: Dict[str, Any]) -> None:
        a = self.arena
        a.flat_p.copy_(snap["p"])
        if a.flat_p16 is not None:
            a.flat_p16.copy_(snap["p16"])  # (whichever shadow arena is current now: both hold pre-step values afterwards)
            if a.flat_p16_alt is not None:
                a.flat_p16_alt.copy_(snap["p16"])
        self.exp_avg.copy_(snap["m"])
        self.exp_avg_sq.copy_(snap["v"])
        self.step_count = snap["step"]

    def state_dict(self) -> Dict[str, Any]:
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq,
                    param_groups=[{k: v for k, v in g.items() if k != "params"} for g in self.param_groups])

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)


class StepInBackward:
    """The optimizer step inside the backward pass: the gradient arena is cut into ranges
    (whole parameters, last-registered first — the order backward produces them); the HIP backward kernels announce every
    parameter gradient they have written (`functional.grad_ready_callbacks`); when the last gradient of a range is in, the
    fused Adam(W) kernel runs over that range on its own stream, ordered after every stream that wrote into it.  The
    reference steps the optimizer after backward (schema.py:977-986 -> optimizers.py:29-33); element for element the
    update is the same kernel on the same inputs, so the parameters are bit-identical to the end-of-step launch
    (tests/test_gpu_train.py).

    Why nothing still running can see a half-updated weight:
      * GEMMs and convolutions read bf16 SHADOWS of the weights; the range updates write the new shadows into a second
        arena (`ParamArena.double_buffer_shadows`) that becomes current when the step is complete;
      * fp32 masters are read during backward only by the normalisation kernels (gamma), and the kernel that reads a
        gamma is the one that writes — and announces — its gradient;
      * a parameter whose gradient arrives through autograd's own accumulation never announces, so its range stays
        open until `launch_step()`; a parameter that announces twice in one pass (shared weights) after its range was
        updated is an error (construct the step engine with `step_in_backward=False` for such models).
    Not combined with global-norm clipping (the norm needs every gradient first) — `FusedAdam.set_step_clip` refuses."""

    accepts_deferred_gradients = True  # acts on explicit notifications only: weight gradients may sit in fused.queue_linear_dw

    def __init__(self, optimizer: FusedAdam, range_bytes: int = 32 << 20):
        from . import functional as HF

        self.opt = optimizer
        a = optimizer.arena
        a.double_buffer_shadows()
        self.ranges: List[List[int]] = []  # [start, end, pending, n_params]
        self.range_of = [0] * len(a.params)
        ids: List[int] = []
        end, acc = a.total, 0
        for i in range(len(a.params) - 1, -1, -1):
            ids.append(i)
            acc += a.params[i].numel() * 4
            if acc >= range_bytes or i == 0:
                for j in ids:
                    self.range_of[j] = len(self.ranges)
                self.ranges.append([a.offsets[i], end, len(ids), len(ids)])
                end, ids, acc = a.offsets[i], [], 0
        self._index = {id(p): i for i, p in enumerate(a.params)}
        self._ready = [False] * len(a.params)
        self._launched = [False] * len(self.ranges)
        self.enabled = True
        self.armed = False  # between prepare_step() and launch_step()
        self.stream = None
        if a.flat_p.is_cuda:
            HF.SideStream.ensure()
            self.stream = HF.SideStream.get(0)  # the weight-gradient lane: most ranges close behind a grouped dW launch on it
        self.launched_in_backward = 0  # ranges of the last step that did not wait for launch_step()
        HF.grad_ready_callbacks.append(self._on_grad)

    def close(self) -> None:
        from . import functional as HF

        if self._on_grad in HF.grad_ready_callbacks:
            HF.grad_ready_callbacks.remove(self._on_grad)

    def begin_step(self) -> None:
        """`FusedAdam.prepare_step()`: the hyper-parameter record of this step is on its way — updates may start.  A backward
        pass outside a prepare_step() / launch_step() pair (somebody only wants gradients) updates nothing."""
        for r in self.ranges:
            r[2] = r[3]
        self._ready = [False] * len(self._ready)
        self._launched = [False] * len(self.ranges)
        self.launched_in_backward = 0
        self.armed = True

    def end_step(self) -> None:
        self.armed = False

    def _on_grad(self, p: Tensor) -> None:
        i = self._index.get(id(p))
        if i is None or not self.enabled or not self.armed or self.stream is None:
            return
        ri = self.range_of[i]
        if self._ready[i]:
            if self._launched[ri]:
                raise RuntimeError(f"parameter #{i} {tuple(p.shape)}: gradient written again after its range was updated "
                                   "(shared weights): build the step engine with step_in_backward=False")
            return
        self._ready[i] = True
        r = self.ranges[ri]
        r[2] -= 1
        if r[2] == 0:
            self._launch(ri)

    def _launch(self, ri: int) -> None:
        from . import functional as HF

        if torch.cuda.is_current_stream_capturing():
            return  # a hipGraph capture records the end-of-step launch instead
        r = self.ranges[ri]
        st = self.stream
        cur = torch.cuda.current_stream()
        if cur != st:
            st.wait_stream(cur)
        for side in HF.SideStream.streams:  # batch slices of the backward, LayerNorm parameter gradients
            if side is not None and side != st:
                st.wait_stream(side)
        self.opt.launch_range(r[0], r[1], st)
        self._launched[ri] = True
        self.launched_in_backward += 1


class FusedAdamOptimizer(torch.optim.Optimizer):
    """`FusedAdam` behind the `torch.optim.Optimizer` interface, for the reference's registries: its schedulers derive
    from `torch.optim.lr_scheduler._LRScheduler` (schedulers.py:13,60,126), which insists on an `Optimizer` instance,
    `accelerate` wraps one, and the DDP callback uses `register_step_pre_hook`.  Same constructor keywords as
    `torch.optim.Adam` / `AdamW` (optimizers.py:29-31: `register_optimizer("adam")(FusedAdamOptimizer)`,
    `register_optimizer("adamw")(FusedAdamWOptimizer)`).  `param_groups` is ONE list shared with the fused optimizer, so
    a scheduler's `group["lr"] = ...` is what the next step's 32-byte hyper-parameter record carries.  One group only:
    the update is one kernel over one flat arena."""

    decoupled_default = False

    def __init__(self, params: Any, lr: float = 1.0e-3, betas: Any = (0.9, 0.999), eps: float = 1.0e-8,
                 weight_decay: float = 0.0, *, decoupled: Optional[bool] = None, arena: Optional[ParamArena] = None):
        if arena is None:
            params = list(params)
            if params and isinstance(params[0], dict):
                if len(params) != 1:
                    raise ValueError("FusedAdamOptimizer: one parameter group only (one kernel over one flat arena)")
                params = list(params[0]["params"])
            arena = ParamArena(params)
        self.fused = FusedAdam(None, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                               decoupled=self.decoupled_default if decoupled is None else decoupled, arena=arena)
        super().__init__(arena.params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.fused.param_groups = self.param_groups  # shared: schedulers write the learning rate here
        self.arena = arena

    def zero_grad(self, set_to_none: bool = False) -> None:  # the arena views stay bound either way
        self.fused.zero_grad()

    def step(self, closure: Any = None) -> Any:  # type: ignore
        loss = None if closure is None else closure()
        self.fused.step()
        return loss

    def state_dict(self) -> Dict[str, Any]:  # type: ignore
        return self.fused.state_dict()

    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:  # type: ignore
        self.fused.load_state_dict(state_dict)


class FusedAdamWOptimizer(FusedAdamOptimizer):
    """Decoupled weight decay (`torch.optim.AdamW`, optimizers.py:31); default weight_decay 1e-2 like torch."""

    decoupled_default = True

    def __init__(self, params: Any, lr: float = 1.0e-3, betas: Any = (0.9, 0.999), eps: float = 1.0e-8,
                 weight_decay: float = 1.0e-2, **kwargs: Any):
        super().__init__(params, lr, betas, eps, weight_decay, **kwargs)


def clip_grad_norm_(arena: ParamArena, max_norm: float, optimizer: Optional[FusedAdam] = None) -> Tensor:
    """Global L2 clipping on the gradient arena (reference trainer.py:170-176: `accelerator.clip_grad_norm_` when
    `clip_norm > 0`).  Returns the norm as a DEVICE scalar and never reads it on the host (accelerate's version
    synchronises every step).  The norm is that of the gradients the update will see: when the arena holds the rank
    SUM and the optimizer applies 1 / W (`grad_scale`), the sum's norm is scaled by it.  With an optimizer the
    coefficient min(1, max_norm / (norm + 1e-6)) — torch's formula — rides in this ONE step's hyper-parameter record
    (`FusedAdam.set_step_clip`); without one the arena is scaled in place."""
    arena.finalize_grads()
    base = float(optimizer.grad_scale) if optimizer is not None else 1.0
    total = ops.sumsq(arena.flat_g).sqrt() * base
    coef = (float(max_norm) / (total + 1.0e-6)).clamp(max=1.0)
    if optimizer is not None:
        optimizer.set_step_clip(coef)
    else:
        arena.flat_g.mul_(coef)
    return total


class EMA(torch.nn.Module):
    """reference modules/common.py:102-162 — same constructor, buffer names (`name.replace(".", "_")`, `num_updates`),
    `forward()` update rule and train() / eval() parameter swapping; the update is ONE kernel over a flat buffer when
    the parameters live in a `ParamArena` (the reference allocates and clones every parameter
    every step — a second full-model HBM pass plus ~150 allocations).  ema = (1 - decay) * p + decay * ema, bit-exact."""

    def __init__(self, decay: float, named_parameters: List[Tuple[str, torch.nn.Parameter]], *,
                 use_num_updates: bool = False, arena: Optional[ParamArena] = None):
        super().__init__()
        self._cache: Dict[str, Tensor] = {}
        self._decay = decay
        self._named_parameters = list(named_parameters)
        params = [p for _, p in self._named_parameters]
        self._arena = arena if (arena is not None and len(arena.params) == len(params)
                                and all(a is b for a, b in zip(arena.params, params))) else None
        if self._arena is not None:
            self._flat = self._arena.flat_p.detach().clone()
            offsets = self._arena.offsets
        else:
            offsets, total = [], 0
            for p in params:
                offsets.append(total)
                total += (p.numel() + 7) // 8 * 8
            dev = params[0].device if params else None
            self._flat = torch.zeros(total, dtype=torch.float32, device=dev)
        for (name, p), off in zip(self.tgt_params, offsets):
            view = self._flat[off:off + p.numel()].view(p.shape)
            if self._arena is None:
                view.copy_(p.data)
            self.register_buffer(name, view)
        self.register_buffer("num_updates", torch.tensor(0 if use_num_updates else -1, dtype=torch.int))
        self._offsets = offsets

    @property
    def tgt_params(self):
        return [(n.replace(".", "_"), p) for n, p in self._named_parameters]

    def _apply(self, fn, *args, **kwargs):  # type: ignore
        """.to() / .cuda(): move the FLAT buffer and re-create the per-parameter views on it"""
        self._flat = fn(self._flat)
        for (name, p), off in zip(self.tgt_params, self._offsets):
            self._buffers[name] = self._flat[off:off + p.numel()].view(p.shape)
        self._buffers["num_updates"] = fn(self._buffers["num_updates"])
        return self

    def forward(self) -> None:
        if not self.training:
            raise ValueError("should not update `EMA` at inference stage")
        if self.num_updates < 0:
            decay = self._decay
        else:
            self.num_updates += 1
            decay = min(self._decay, (1 + int(self.num_updates)) / (10 + int(self.num_updates)))
        if self._arena is not None:
            ops.ema_update(self._flat, self._arena.flat_p, decay)
            return
        for (name, p), off in zip(self.tgt_params, self._offsets):
            n8 = (p.numel() + 7) // 8 * 8
            if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.data_ptr() % 16 == 0:
                ops.ema_update(self._flat[off:off + p.numel()], p.data.view(-1), decay)
            else:
                raise RuntimeError(f"EMA: parameter '{name}' must be a contiguous, 16-byte aligned f32 device tensor "
                                   f"(slot {off}..{off + n8}); put the parameters in a ParamArena")

    def train(self, mode: bool = True) -> "EMA":
        super().train(mode)
        if mode:
            for name, param in self.tgt_params:
                cached = self._cache.pop(name, None)
                if cached is not None:
                    param.data.copy_(cached)
        else:
            for name, param in self.tgt_params:
                if name not in self._cache:
                    self._cache[name] = param.data.clone()
                param.data.copy_(getattr(self, name))
        # the forward kernels read bf16 copies of the weights, cached on `param._version` — which `.data.copy_()` does
        # not bump: refresh the arena's shadows in one pass, invalidate every other cached shadow
        if self._arena is not None and self._arena.flat_p.is_cuda:
            self._arena.refresh_shadow()
        else:
            for _, param in self.tgt_params:
                if hasattr(param, "_cfhip_shadow_version"):
                    param._cfhip_shadow_version = None
        return self
