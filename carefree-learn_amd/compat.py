"""Seams for users of the reference package (`cflearn`) itself — optional patches applied to the IMPORTED reference,
never to its source tree.

`patch_lazy_losses()` is SURVEY §8(f) rank 3 for cflearn users: the reference's `CommonTrainStep.loss_fn`
(models/common.py:31-43) calls `.item()` on every loss key EVERY step — one host synchronisation per key per step, which
at MI355X step times (a ViT-B/16 step is ~22 ms, an FCNN step tens of microseconds) serialises the host with the GPU.
The patch keeps the loss tensors on the device and hands the trainer `LazyFloat`s that synchronise only when somebody
actually reads them (the monitor / logging steps: `trainer.py:555-564`, `callbacks/general.py:200`), which is every
`num_step_per_snapshot` steps instead of every step.
"""
from typing import Any, Dict


class LazyFloat:
    """A device scalar that behaves like the float the reference expects, read back on first use."""

    __slots__ = ("_t", "_v")

    def __init__(self, t: Any):
        self._t, self._v = t.detach(), None

    def item(self) -> float:
        if self._v is None:
            self._v = float(self._t)  # the one host synchronisation, only when the value is consumed
            self._t = None
        return self._v

    __float__ = item

    @property
    def is_materialized(self) -> bool:
        return self._v is not None

    def __format__(self, spec: str) -> str:
        return format(self.item(), spec)

    def __repr__(self) -> str:
        return repr(self.item()) if self._v is not None else "LazyFloat(<on device>)"

    # arithmetic / comparisons the reference applies to loss items (schema.py:989-1005: negation, sums, weights)
    def __neg__(self) -> float:
        return -self.item()

    def __add__(self, o: Any) -> float:
        return self.item() + float(o)

    __radd__ = __add__

    def __sub__(self, o: Any) -> float:
        return self.item() - float(o)

    def __rsub__(self, o: Any) -> float:
        return float(o) - self.item()

    def __mul__(self, o: Any) -> float:
        return self.item() * float(o)

    __rmul__ = __mul__

    def __truediv__(self, o: Any) -> float:
        return self.item() / float(o)

    def __lt__(self, o: Any) -> bool:
        return self.item() < float(o)

    def __gt__(self, o: Any) -> bool:
        return self.item() > float(o)

    def __eq__(self, o: Any) -> bool:  # type: ignore
        return self.item() == float(o)

    def __hash__(self) -> int:
        return hash(self.item())


def patch_lazy_losses(cflearn: Any = None) -> Any:
    """Replace `cflearn.models.common.CommonTrainStep.loss_fn` by a version without the per-step `.item()`.
    Returns the original method (call `restore_losses(orig)` to undo)."""
    import importlib

    common = importlib.import_module("cflearn.models.common")
    constants = importlib.import_module("cflearn.constants")
    orig = common.CommonTrainStep.loss_fn

    def loss_fn(self: Any, m: Any, state: Any, batch: Dict[str, Any], forward_results: Any, **kwargs: Any) -> Any:
        losses = self.loss.run(forward_results, batch, state)
        return common.TrainStepLoss(losses[constants.LOSS_KEY], {k: LazyFloat(v) for k, v in losses.items()})

    common.CommonTrainStep.loss_fn = loss_fn
    return orig


def restore_losses(orig: Any) -> None:
    import importlib

    importlib.import_module("cflearn.models.common").CommonTrainStep.loss_fn = orig
