"""Input path: `cflearn_amd.data.TensorBatcher` keeps the reference batcher's interface and conversion rules
(data/utils.py:255-283, toolkit.py:1182-1207) — CPU mode here, pinned / copy-stream mode in the gpu-marked test."""
import numpy as np
import pytest
import torch

from cflearn_amd.data import TensorBatcher


class _Loader:
    def __init__(self, n, bs=4, last=None):
        rng = np.random.default_rng(0)
        sizes = [bs] * (n - 1) + [last or bs]
        self.batches = [dict(input=rng.standard_normal((s, 3, 8, 8)), labels=rng.integers(0, 10, (s, 1), dtype=np.int32),
                             names=np.array([f"s{i}" for i in range(s)])) for s in sizes]

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)

    def get_one_batch(self):
        return self.batches[0]

    def get_full_batch(self):
        return self.batches[0]


def _check(batcher, loader, device):
    assert len(batcher) == len(loader)
    for epoch in range(2):  # re-iterable, like the reference
        count = 0
        for b in batcher:  # compared as yielded: a batch is valid until the next one is requested (buffer ring)
            ref = loader.batches[count]
            count += 1
            assert set(b) == set(ref)
            assert b["input"].dtype == torch.float32 and b["labels"].dtype == torch.int64
            assert b["input"].device.type == device and b["labels"].device.type == device
            assert torch.equal(b["input"].cpu(), torch.from_numpy(ref["input"].astype(np.float32)))
            assert torch.equal(b["labels"].cpu(), torch.from_numpy(ref["labels"].astype(np.int64)))
            assert b["names"] is ref["names"]  # strings untouched
        assert count == len(loader.batches)
    one = batcher.get_one_batch()
    assert one["input"].device.type == device and one["input"].shape[0] == loader.batches[0]["input"].shape[0]


def test_tensor_batcher_cpu_semantics():
    loader = _Loader(5, last=3)
    _check(TensorBatcher(loader, None), loader, "cpu")


@pytest.mark.gpu
def test_tensor_batcher_prefetches_bit_exact_on_gpu():
    loader = _Loader(7, last=2)
    _check(TensorBatcher(loader, 0, depth=3), loader, "cuda")
    _check(TensorBatcher(loader, "cuda:0", depth=1), loader, "cuda")
