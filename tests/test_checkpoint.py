"""AsyncCheckpointer: the reference's checkpoint files (trainer.py:380-419, schema.py:1377-1382) written off the
training thread."""
import json
import os

import pytest
import torch

from cflearn_amd.checkpoint import PT_PREFIX, SCORES_FILE, AsyncCheckpointer, get_sorted_checkpoints


def _model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.BatchNorm1d(8), torch.nn.Linear(8, 2))


def test_snapshot_semantics_format_and_topk_prune(tmp_path):
    m = _model()
    ck = AsyncCheckpointer(str(tmp_path), max_snapshot_file=2)
    want = {}
    for step, score in ((10, 0.3), (20, 0.9), (30, 0.5), (40, 0.1)):
        want[step] = {k: v.clone() for k, v in m.state_dict().items()}
        name = ck.save(step, score, m.state_dict(), config=dict(module_name="fcnn"))
        assert name == f"{PT_PREFIX}{step}.pt"
        with torch.no_grad():  # training goes on immediately: the snapshot must not see this
            for p in m.parameters():
                p.add_(1.0)
    ck.close()
    # the reference's rule, applied before every save: keep the best (max - 1), then add the new one
    #   10 -> {10}; 20 -> {20, 10}... before 30: [20, 10] -> drop 10; before 40: [20, 30] -> drop 30
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".pt"))
    assert files == [f"{PT_PREFIX}20.pt", f"{PT_PREFIX}40.pt"]
    scores = json.load(open(tmp_path / SCORES_FILE))
    assert list(scores.items()) == [(f"{PT_PREFIX}20.pt", 0.9), (f"{PT_PREFIX}40.pt", 0.1)]  # best first
    assert get_sorted_checkpoints(str(tmp_path)) == [f"{PT_PREFIX}20.pt", f"{PT_PREFIX}40.pt"]
    for step in (20, 40):
        full = torch.load(tmp_path / f"{PT_PREFIX}{step}.pt")
        assert set(full) == {"config", "states"} and full["config"] == dict(module_name="fcnn")
        assert list(full["states"].keys()) == list(want[step].keys())
        for k, v in want[step].items():
            assert torch.equal(full["states"][k], v), (step, k)
    # a restored model is the model as of the call
    m2 = _model()
    m2.load_state_dict(torch.load(tmp_path / f"{PT_PREFIX}40.pt")["states"])
    for k, v in want[40].items():
        assert torch.equal(m2.state_dict()[k], v)


def test_resume_keeps_history_and_errors_surface(tmp_path):
    m = _model()
    ck = AsyncCheckpointer(str(tmp_path), max_snapshot_file=3)
    ck.save(1, 0.2, m.state_dict())
    ck.close()
    ck = AsyncCheckpointer(str(tmp_path), max_snapshot_file=3)  # a resumed run sees the earlier scores
    ck.save(2, 0.4, m.state_dict())
    ck.wait()
    assert get_sorted_checkpoints(str(tmp_path)) == [f"{PT_PREFIX}2.pt", f"{PT_PREFIX}1.pt"]
    ck.save(3, 0.1, {"bad": (lambda: 0)})  # not picklable: the writer thread fails ...
    with pytest.raises(RuntimeError, match="checkpoint write failed"):
        ck.wait()  # ... and the training thread hears about it
    ck.close()


@pytest.mark.gpu
def test_device_snapshot_is_taken_at_call_time(tmp_path):
    dev = torch.device("cuda")
    m = _model().to(dev)
    ck = AsyncCheckpointer(str(tmp_path), max_snapshot_file=0)
    want = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ck.save(7, 1.0, m.state_dict())
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(0.0)  # enqueued after the snapshot copies on the compute stream
    ck.close()
    got = torch.load(tmp_path / f"{PT_PREFIX}7.pt")["states"]
    for k, v in want.items():
        assert torch.equal(got[k], v), k
