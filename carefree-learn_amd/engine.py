"""One optimisation step of a classifier on the HIP path, optionally replayed from a hipGraph.

This is the MI355X-side equivalent of what the reference's step engine does per batch
(`IDLModel.train`, schema.py:1174-1294: forward -> loss -> `accelerator.backward` ->
`optimizer.step` -> `zero_grad`), minus its per-step host synchronisations (`.item()` per loss
key, models/common.py:40-43): the loss stays on the device until somebody asks for it.

Single GPU: the whole step (zero-grad marking, forward, softmax-CE, backward, fused Adam) is
captured once into a hipGraph (through torch's CUDAGraph plumbing — our kernels launch on torch's
current stream, so the capture picks them up) and replayed, which removes ~300 kernel-launch
round trips of Python / ctypes overhead per step.  Only the 32-byte Adam hyper-parameter record is
refreshed from the host before each replay.

Multi GPU (one process per GPU): eager launches; every gradient bucket is all-reduced on the side
stream as soon as the backward kernels have produced it (`ddp.BucketedAllReduce`), the optimizer
waits for the events, 1/W is folded into the Adam kernel.
"""
from typing import Any, Dict, Optional

import torch
from torch import Tensor

from . import ops
from .functional import SideStream
from .ddp import BucketedAllReduce
from .optim import FusedAdam, ParamArena, clip_grad_norm_


class TrainStep:
    def __init__(self, model: torch.nn.Module, *, lr: float = 1.0e-4, betas: Any = (0.9, 0.999),
                 eps: float = 1.0e-8, weight_decay: float = 0.0, decoupled: bool = True,
                 use_graph: bool = False, distributed: bool = False, bucket_bytes: int = 64 << 20,
                 output_key: str = "predictions", loss: str = "cross_entropy", clip_norm: float = 0.0,
                 wire_bf16: bool = False, comm: str = "torch", step_in_backward: bool = True, range_bytes: int = 32 << 20):
        if loss not in ("cross_entropy", "focal"):
            raise ValueError(f"unknown loss '{loss}' (cross_entropy: losses/basic.py:126-141, focal: :170-206)")
        self.model = model
        self.output_key = output_key
        self.loss = loss
        self.clip_norm = float(clip_norm)  # reference TrainerConfig.clip_norm (trainer.py:170-176); 0 = off
        self.grad_norm: Optional[Tensor] = None  # device scalar of the last clipped step
        params = [p for p in model.parameters() if p.requires_grad]
        self.arena = ParamArena(params, with_shadow=True)
        self.optimizer = FusedAdam(None, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                   decoupled=decoupled, arena=self.arena)
        self.optimizer.lazy_zero = True  # every parameter gradient is written by a HIP backward kernel
        # the optimizer update rides inside backward, range by range (optim.StepInBackward); clipping needs the global norm
        # first and a hipGraph records the one end-of-step launch
        in_bwd = bool(step_in_backward) and self.clip_norm == 0.0 and not use_graph and bool(params) and params[0].is_cuda
        if params and params[0].is_cuda:
            SideStream.ensure()  # the stream self-check runs here, not inside the first timed step
        self.reducer: Optional[BucketedAllReduce] = None
        if distributed:
            communicator = None
            if comm == "cfhip":  # collectives through the cfhip_comm_* C-ABI on this package's own comm stream
                from .ddp import Communicator

                communicator = Communicator()
            elif comm != "torch":
                raise ValueError(f"comm = '{comm}': 'torch' (torch.distributed launches the collectives) or 'cfhip'")
            self.reducer = BucketedAllReduce(self.arena, bucket_bytes=bucket_bytes, optimizer=self.optimizer,
                                             wire_bf16=wire_bf16, comm=communicator, step_in_backward=in_bwd)
            self.reducer.broadcast_parameters(0)
        elif in_bwd:
            self.optimizer.enable_step_in_backward(range_bytes)
        self.use_graph = use_graph and not distributed
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._static: Dict[str, Tensor] = {}
        self.loss_sum: Optional[Tensor] = None  # device scalar (sum over the local batch)

    # -- the step body (all device work, no host sync) ---------------------------------------------
    def _body(self, img: Tensor, labels: Tensor) -> Tensor:
        self.optimizer.zero_grad()
        out = self.model(img)
        logits = out[self.output_key] if isinstance(out, dict) else out
        if self.loss == "focal":
            loss_sum, dlogits = ops.softmax_focal(logits, labels, 1.0 / logits.shape[0])
        else:
            loss_sum, dlogits = ops.softmax_xent(logits, labels, 1.0 / logits.shape[0])
        logits.backward(dlogits)
        SideStream.join()  # parameter-gradient kernels ran on the side stream
        if self.reducer is not None:
            self.reducer.finish()
        if self.clip_norm > 0.0:
            self.grad_norm = clip_grad_norm_(self.arena, self.clip_norm, self.optimizer)  # device-side, no host read
        self.optimizer.launch_step()
        return loss_sum

    def _capture(self, img: Tensor, labels: Tensor) -> None:
        self._static = dict(img=img.clone(), labels=labels.clone())
        snap = self.optimizer.snapshot()  # the warm-up passes are not training steps: their updates are rolled back
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up off the capture stream (allocator, lazy inits)
            for _ in range(2):
                self.optimizer.prepare_step()
                self._body(self._static["img"], self._static["labels"])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.optimizer.restore(snap)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):  # records, does not execute
            self.loss_sum = self._body(self._static["img"], self._static["labels"])

    def step(self, img: Tensor, labels: Tensor) -> Tensor:
        """Runs one step; returns the device tensor holding the summed CE loss of the batch."""
        if self.use_graph:
            if self._graph is None:
                self._capture(img, labels)  # 2 eager warm-up passes (rolled back), then the recording
            if img.data_ptr() != self._static["img"].data_ptr():
                self._static["img"].copy_(img, non_blocking=True)
                self._static["labels"].copy_(labels, non_blocking=True)
            self.optimizer.prepare_step()
            self._graph.replay()
            return self.loss_sum.clone()  # (the recorded tensor is overwritten by every replay)
        self.optimizer.prepare_step()
        self.loss_sum = self._body(img, labels)
        return self.loss_sum


class LossTrainStep:
    """The same step engine for any module whose loss is computed by HIP autograd functions: `loss_fn(model, batch)`
    returns a device scalar (e.g. `CLIP.contrastive_loss`); backward; side-stream join; bucketed gradient all-reduce
    when `distributed`; fused Adam(W) over the arena.  No host synchronisation inside `step()`."""

    def __init__(self, model: torch.nn.Module, loss_fn: Any, *, lr: float = 1.0e-4, betas: Any = (0.9, 0.999),
                 eps: float = 1.0e-8, weight_decay: float = 0.0, decoupled: bool = True, distributed: bool = False,
                 bucket_bytes: int = 64 << 20, step_in_backward: bool = True, range_bytes: int = 32 << 20):
        self.model, self.loss_fn = model, loss_fn
        params = [p for p in model.parameters() if p.requires_grad]
        self.arena = ParamArena(params, with_shadow=True)
        self.optimizer = FusedAdam(None, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled,
                                   arena=self.arena)
        # NOT lazy: parameters such as CLIP's logit_scale receive their gradient through autograd's accumulation
        self.optimizer.lazy_zero = False
        if params and params[0].is_cuda:
            SideStream.ensure()
        self.reducer: Optional[BucketedAllReduce] = None
        # the update of an arena range runs inside backward once its last gradient is announced (optim.StepInBackward);
        # parameters whose gradient comes through autograd's accumulation (logit_scale) never announce: their range waits
        # for launch_step()
        in_bwd = bool(step_in_backward) and bool(params) and params[0].is_cuda
        if distributed:
            self.reducer = BucketedAllReduce(self.arena, bucket_bytes=bucket_bytes, optimizer=self.optimizer, step_in_backward=in_bwd)
            self.reducer.broadcast_parameters(0)
        elif in_bwd:
            self.optimizer.enable_step_in_backward(range_bytes)
        self.loss: Optional[Tensor] = None

    def step(self, batch: Any) -> Tensor:
        self.optimizer.prepare_step()
        self.optimizer.zero_grad()
        self.loss = self.loss_fn(self.model, batch)
        self.loss.sum().backward()
        SideStream.join()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.launch_step()
        return self.loss
