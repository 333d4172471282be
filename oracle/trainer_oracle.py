"""CPU restatement of the reference's STEP ENGINE — what runs once per batch between the data loader and the
optimizer.  TEST INFRASTRUCTURE: only tests/, smoke() and bench.py's cpu_baseline may
import this; the product never does.

It exists because the reference tree is absent on the GPU box: the GPU tests drive this repo's HIP modules, the
`RcclDDPCallback` seam and the fused optimizers through THIS loop, and `tests/test_reference_engine.py` pins the loop
against the reference's own `IDLModel.train` / `get_update_fn` / `Trainer.clip_norm_step` on CPU (same model, same
batches: bit-equal losses and weights), with the reference imported from /root/reference through oracle/refharness.

Restated (reference file:line, carefree-learn v0.5.0):
  * `Trainer.fit` inner loop ................ trainer.py:312-347  (before_loop callbacks, `state.step += 1`, `_step`,
                                               after_step callbacks)
  * `Trainer._step` .......................... trainer.py:579-587  (`to_device`, forward / loss kwargs, `model.train`)
  * `IDLModel.train`, one train step ........ schema.py:1239-1294 (forward under autocast, loss, update rule
                                               `state.step % grad_accumulate == 0`, scheduler step)
  * `IDLModel.run` / `postprocess` .......... schema.py:1398-1408,1123-1137 (`m(batch["input"])`, tensor -> {"predictions"})
  * `get_update_fn` .......................... schema.py:977-986   (backward; if update: clip, step, zero_grad)
  * `Trainer.clip_norm_step` ................. trainer.py:170-176  (`clip_grad_norm_` iff clip_norm > 0)
  * `CommonTrainStep.loss_fn` ................ models/common.py:31-43 (`loss.run`, `.item()` per loss key: a host sync)
  * `ILoss.run` / `_reduce` .................. schema.py:757-807   (predictions + labels, mean reduction)
  * `FocalLoss` / `CrossEntropyLoss` ......... losses/basic.py:126-141,170-206
"""
from typing import Any, Callable, Dict, Iterable, List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

LOSS_KEY, INPUT_KEY, LABEL_KEY, PREDICTIONS_KEY = "loss", "input", "labels", "predictions"  # constants.py:3-7


# -- losses (losses/basic.py) --------------------------------------------------------------------------------------


def cross_entropy_losses(predictions: Tensor, labels: Tensor) -> Tensor:
    """losses/basic.py:126-141: -log_softmax(pred).gather(1, labels); labels int64 [B, 1]; un-reduced [B, 1]"""
    return -F.log_softmax(predictions, dim=1).gather(dim=1, index=labels)


def focal_losses(predictions: Tensor, labels: Tensor, eps: float = 1.0e-6, gamma: float = 2.0) -> Tensor:
    """losses/basic.py:170-206 with input_logits=True, alpha=None"""
    prob_mat = F.softmax(predictions.view(-1, predictions.shape[-1]), dim=1) + eps
    p = prob_mat.gather(dim=1, index=labels).view(-1)
    return (-p.log() * (1 - p) ** gamma).view_as(labels)


LOSSES: Dict[str, Callable[[Tensor, Tensor], Tensor]] = {"cross_entropy": cross_entropy_losses, "focal": focal_losses}


def run_loss(name: str, forward_results: Dict[str, Tensor], batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """ILoss.run (schema.py:798-807): forward on (predictions, labels), then `postprocess` = mean reduction under
    the key "loss"."""
    return {LOSS_KEY: LOSSES[name](forward_results[PREDICTIONS_KEY], batch[LABEL_KEY]).mean()}


# -- the engine ------------------------------------------------------------------------------------------------------


class State:
    """TrainerState (schema.py:1535-1603): only the counters the step engine reads."""

    def __init__(self) -> None:
        self.step = self.epoch = 0


class Config:
    """The TrainerConfig fields the step engine reads (schema.py:1876-1949)."""

    def __init__(self, grad_accumulate: int = 1, clip_norm: float = 0.0, mixed_precision: str = "no"):
        self.grad_accumulate, self.clip_norm, self.mixed_precision = grad_accumulate, clip_norm, mixed_precision


class _Step:
    """TrainStep (schema.py:1016-1062) of scope "all" with the defaults CommonTrainStep uses."""

    scope = "all"
    grad_accumulate: Optional[int] = None


class _Model:
    def __init__(self, m: torch.nn.Module):
        self.m = m
        self.train_steps = [_Step()]


class StepEngine:
    """Duck-types what `RcclDDPCallback.before_loop(trainer)` and friends touch on the reference trainer:
    `.model.m`, `.model.train_steps`, `.optimizers`, `.state`, `.config`."""

    def __init__(self, module: torch.nn.Module, loss_name: str, optimizer: Any, *, grad_accumulate: int = 1,
                 clip_norm: float = 0.0, callbacks: Iterable[Any] = (), scheduler: Any = None,
                 lazy_losses: bool = False, device: Any = None):
        self.model = _Model(module)
        self.loss_name = loss_name
        self.optimizers = {"all": optimizer}
        self.scheduler = scheduler
        self.state = State()
        self.config = Config(grad_accumulate, clip_norm)
        self.callbacks = list(callbacks)
        self.lazy_losses = lazy_losses  # the (f)3 seam: keep the loss tensors, read them only when somebody asks
        self.device = device
        self.gradient_norm: Any = None
        self.loss_log: List[Any] = []

    # trainer.py:170-176
    def clip_norm_step(self) -> None:
        if self.config.clip_norm > 0.0:
            params = [p for p in self.model.m.parameters() if p.requires_grad]
            self.gradient_norm = torch.nn.utils.clip_grad_norm_(params, max_norm=self.config.clip_norm)

    # schema.py:977-986
    def _update(self, loss: Tensor, optimizer: Any, update: bool) -> None:
        loss.backward()  # accelerator.backward(loss) without a scaler
        if update:
            self.clip_norm_step()
            optimizer.step()
            optimizer.zero_grad()

    # schema.py:1239-1294 for ONE train step (CommonDLModel.train_steps, models/common.py:50-52)
    def train_step(self, batch_idx: int, batch: Dict[str, Tensor]) -> Dict[str, Any]:
        forward = self.model.m(batch[INPUT_KEY])  # IDLModel.run -> forward(*get_forward_args)
        if isinstance(forward, Tensor):           # IDLModel.postprocess
            forward = {PREDICTIONS_KEY: forward}
        losses = run_loss(self.loss_name, forward, batch)
        if self.lazy_losses:
            loss_items: Dict[str, Any] = {k: v.detach() for k, v in losses.items()}
        else:
            loss_items = {k: v.item() for k, v in losses.items()}  # models/common.py:40-43: one host sync per key
        step = self.model.train_steps[0]
        update = self.state.step % (step.grad_accumulate or self.config.grad_accumulate) == 0
        self._update(losses[LOSS_KEY], self.optimizers[step.scope], update)
        if update and self.scheduler is not None:
            self.scheduler.step()  # trainer.scheduler_step()
        return loss_items

    # trainer.py:312-347 (no monitors / checkpoints / tqdm: SURVEY §8d keeps those out of the timed window anyway)
    def fit(self, batches: Iterable[Dict[str, Tensor]]) -> "StepEngine":
        for cb in self.callbacks:
            cb.before_loop(self)
        for i, batch in enumerate(batches):
            self.state.step += 1
            if self.device is not None:
                batch = {k: v.to(self.device) for k, v in batch.items()}  # to_device(batch, self.device)
            self.loss_log.append(self.train_step(i, batch))
            for cb in self.callbacks:
                after = getattr(cb, "after_step", None)
                if after is not None:
                    after(self.loss_log[-1], self.state)
        return self
