"""Asynchronous snapshots in the reference's checkpoint format.

The reference saves on rank 0 inside the step loop (`Trainer.save_checkpoint`, trainer.py:380-419): `model.save` =
`torch.save(dict(config=..., states=state_dict), "model_{step}.pt")` (schema.py:1377-1382), then `scores.json`
(`{file: score}` sorted by score, best first) and a top-k prune by score (constants.py:11-13, trainer.py:58-71,405-419).
That is a device-to-host copy of every parameter plus pickling plus a disk write, all on the training thread.

Here `save()` only ENQUEUES: the state is copied device -> pinned host staging on a copy stream (ordered after the
compute stream, so it is the state as of the call), an event marks the end of the copies, and a writer thread waits
for that event, prunes, writes the `.pt` file and rewrites `scores.json` — same file names, same payload layout, same
pruning rule, so `Trainer.restore_checkpoint` / `IDLModel.load` read the result unchanged.  Training continues at once;
two staging sets alternate, a third pending snapshot waits for the oldest to reach the disk.
"""
import json
import os
import queue
import threading
from typing import Any, Callable, Dict, List, Optional

import torch
from torch import Tensor

PT_PREFIX = "model_"          # reference constants.py:11
SCORES_FILE = "scores.json"   # reference constants.py:12


def get_scores(folder: str) -> Dict[str, float]:
    """reference trainer.py:58-63"""
    path = os.path.join(folder, SCORES_FILE)
    if not os.path.isfile(path):
        return {}
    with open(path, "r") as f:
        return json.load(f)


def _sorted_desc(d: Dict[str, float]) -> Dict[str, float]:
    # cftool's sort_dict_by_value(reverse=True): a stable sort of the items by value
    return dict(sorted(d.items(), key=lambda kv: kv[1], reverse=True))


def get_sorted_checkpoints(folder: str) -> List[str]:
    """Best first (reference trainer.py:66-72)."""
    return list(_sorted_desc(get_scores(folder)).keys())


class _StagingSet:
    def __init__(self) -> None:
        self.buffers: Dict[str, Tensor] = {}
        self.event: Optional[Any] = None
        self.done = threading.Event()
        self.done.set()

    def buffer_for(self, name: str, t: Tensor) -> Tensor:
        b = self.buffers.get(name)
        if b is None or b.shape != t.shape or b.dtype != t.dtype:
            b = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=t.is_cuda)
            self.buffers[name] = b
        return b


class AsyncCheckpointer:
    def __init__(self, folder: str, *, max_snapshot_file: int = 5):
        self.folder = folder
        self.max_snapshot_file = max_snapshot_file
        os.makedirs(folder, exist_ok=True)
        self.checkpoint_scores: Dict[str, float] = get_scores(folder)
        self._sets = [_StagingSet(), _StagingSet()]
        self._turn = 0
        self._copy_stream: Optional[Any] = None
        self._q: "queue.Queue[Optional[Callable[[], None]]]" = queue.Queue()
        self._error: Optional[BaseException] = None
        self._thread = threading.Thread(target=self._run, name="cfhip-checkpoint-writer", daemon=True)
        self._thread.start()

    # -- writer thread ---------------------------------------------------------------------------------------------
    def _run(self) -> None:
        while True:
            job = self._q.get()
            if job is None:
                return
            try:
                job()
            except BaseException as e:  # surfaced by the next save() / wait()
                self._error = e

    def _raise_pending(self) -> None:
        if self._error is not None:
            e, self._error = self._error, None
            raise RuntimeError("asynchronous checkpoint write failed") from e

    # -- training thread -------------------------------------------------------------------------------------------
    def save(self, step: int, score: float, states: Dict[str, Tensor], config: Optional[Dict[str, Any]] = None, *,
             no_history: bool = False) -> str:
        """Snapshot `states` (a state_dict: tensors on the HIP device or the host) as of NOW and write it in the
        background.  Returns the file name (`model_{step}.pt`)."""
        self._raise_pending()
        st = self._sets[self._turn]
        self._turn ^= 1
        st.done.wait()  # its previous snapshot has reached the disk
        st.done.clear()
        on_device = any(t.is_cuda for t in states.values() if isinstance(t, Tensor))
        host: Dict[str, Any] = {}
        if on_device:
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream()
            self._copy_stream.wait_stream(torch.cuda.current_stream())  # everything enqueued so far is in the snapshot
            with torch.cuda.stream(self._copy_stream):
                for k, t in states.items():
                    if isinstance(t, Tensor):
                        t = t.detach()
                        if t.is_cuda:
                            b = st.buffer_for(k, t)
                            b.copy_(t, non_blocking=True)
                            t.record_stream(self._copy_stream)
                            host[k] = b
                        else:
                            host[k] = t.clone()
                    else:
                        host[k] = t
                st.event = torch.cuda.Event()
                st.event.record()
        else:
            for k, t in states.items():
                host[k] = t.detach().clone() if isinstance(t, Tensor) else t
            st.event = None
        pt_file = f"{PT_PREFIX}{step}.pt"
        self._q.put(lambda: self._write(st, pt_file, float(score), host, config, no_history))
        return pt_file

    def _write(self, st: _StagingSet, pt_file: str, score: float, host: Dict[str, Any],
               config: Optional[Dict[str, Any]], no_history: bool) -> None:
        try:
            if st.event is not None:
                st.event.synchronize()
            # leave top_k snapshots only (reference trainer.py:405-411)
            if self.max_snapshot_file > 0:
                checkpoints = get_sorted_checkpoints(self.folder)
                if len(checkpoints) >= self.max_snapshot_file:
                    for file in checkpoints[self.max_snapshot_file - 1:]:
                        self.checkpoint_scores.pop(file, None)
                        path = os.path.join(self.folder, file)
                        if os.path.isfile(path):
                            os.remove(path)
            # pt: the payload of IDLModel.save (reference schema.py:1377-1382); written under a temporary name first so
            # that a reader never sees a half-written snapshot
            full = dict(config=config if config is not None else {}, states=host)
            tmp = os.path.join(self.folder, pt_file + ".tmp")
            torch.save(full, tmp)
            os.replace(tmp, os.path.join(self.folder, pt_file))
            # scores (reference trainer.py:414-419)
            scores = {} if no_history else self.checkpoint_scores
            scores[pt_file] = score
            with open(os.path.join(self.folder, SCORES_FILE), "w") as f:
                json.dump(_sorted_desc(scores), f)
        finally:
            st.done.set()

    def wait(self) -> None:
        """Block until every snapshot enqueued so far is on the disk."""
        for st in self._sets:
            st.done.wait()
        self._raise_pending()

    def close(self) -> None:
        self.wait()
        self._q.put(None)
        self._thread.join(timeout=60)
