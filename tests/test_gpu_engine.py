"""The reference's STEP ENGINE over this package's HIP modules (boundary row b).

`/root/reference` is absent on the GPU box, so the loop is `oracle/trainer_oracle.StepEngine` — the restatement of
`Trainer.fit` / `IDLModel.train` / `get_update_fn` / `clip_norm_step` that `tests/test_reference_engine.py` pins
bit-for-bit against the reference's own code on CPU.  It drives: modules built through the registry
(`build_module("cv_clf", ...)`), `FusedAdamWOptimizer` (registered where the reference registers torch's AdamW),
`RcclDDPCallback.before_loop` (the trainer-side seam, 1-rank RCCL process group) and, for (f)3, the lazy-loss seam.
The result must be the trajectory of this package's own engine (`engine.TrainStep`)."""
import os

import pytest
import torch
import torch.distributed as dist

import trainer_oracle as TO
import cflearn_amd as C
from cflearn_amd.engine import TrainStep
from cflearn_amd.optim import FusedAdamWOptimizer
from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _small(golden):
    g = golden("vit_small.pt")
    cfg = dict(g["cfg"])
    m = C.build_module("cv_clf", config=dict(in_channels=3, num_classes=g["num_classes"], img_size=cfg.pop("img_size"),
                                             latent_dim=cfg["latent_dim"], encoder="vit", encoder_config=cfg))
    m.load_state_dict(g["sd"])
    return g, m.to(DEV)


@pytest.fixture(scope="module")
def one_rank_rccl():
    import socket

    created = False
    if not dist.is_initialized():
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        saved = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dev0 = torch.device("cuda", 0)
        torch.cuda.set_device(dev0)
        dist.init_process_group("nccl", device_id=dev0, rank=0, world_size=1)
        created = True
    yield
    if created:
        dist.destroy_process_group()
        for k, v in saved.items():  # `get_ddp_info()` reads these: do not leak a "distributed" environment
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("lazy,clip_norm,grad_accumulate", [(False, 0.0, 1), (True, 0.0, 1), (True, 0.5, 1), (False, 0.0, 2)])
def test_reference_step_engine_drives_the_hip_path(golden, one_rank_rccl, lazy, clip_norm, grad_accumulate):
    lr, wd, steps = 1e-3, 0.0, 4
    g, ref_model = _small(golden)
    img = g["img"].to(DEV)
    labels = g["labels"].view(-1, 1).to(DEV)  # the reference's label layout: int64 [B, 1]

    # (1) this package's own engine: one step per `grad_accumulate` micro-batches is not its business, so feed it the
    # accumulated batch semantics directly when grad_accumulate == 1 only
    own_losses = []
    if grad_accumulate == 1:
        ts = TrainStep(ref_model, lr=lr, weight_decay=wd, decoupled=True, clip_norm=clip_norm)
        for _ in range(steps):
            own_losses.append(ts.step(img, labels.view(-1)).item() / img.shape[0])

    # (2) the reference's step engine (restated) over the same modules
    _, model = _small(golden)
    opt = FusedAdamWOptimizer(model.parameters(), lr=lr, weight_decay=wd)
    cb = C.RcclDDPCallback(bucket_bytes=1 << 20)
    eng = TO.StepEngine(model, "cross_entropy", opt, callbacks=[cb], lazy_losses=lazy, clip_norm=clip_norm,
                        grad_accumulate=grad_accumulate)
    batch = {TO.INPUT_KEY: img, TO.LABEL_KEY: labels}
    eng.fit([batch] * steps)
    assert cb.reducer is not None and cb.reducer.arena is opt.arena  # the seam was installed on the fused optimizer's arena
    # round 3: under RCCL the callback owns a C-ABI communicator (cfhip_comm_*): its collectives run on this package's own,
    # queue-checked comm stream, not on the ProcessGroup's internal one
    assert cb.reducer.comm is not None and cb.reducer.comm.world == 1
    assert cb.reducer.comm.count() == (1, 0)  # ncclCommCount / ncclCommUserRank: what RCCL says, not the Python bookkeeping
    # round 4: the stream plan fits the four hardware queues WITH a ProcessGroup alive — caller's stream,
    # two side lanes, comm stream: every helper stream passed its own-queue check (round 3 warned here in all four cases)
    from cflearn_amd import functional as HF

    rep = HF.stream_report()
    assert rep["distinct"], rep
    assert HF.SideStream.lanes == 2 and cb.reducer.comm_stream is not None
    assert HF._overlap([torch.cuda.current_stream(), cb.reducer.comm_stream] + [st for st in HF.SideStream.streams if st is not None])
    got = [float(d[TO.LOSS_KEY]) for d in eng.loss_log]
    assert all(torch.isfinite(torch.tensor(got)))
    if lazy:
        assert all(isinstance(d[TO.LOSS_KEY], torch.Tensor) and d[TO.LOSS_KEY].is_cuda for d in eng.loss_log)
    if grad_accumulate == 1:
        for a, b in zip(got, own_losses):
            assert abs(a - b) <= 2e-3 * abs(b), (got, own_losses)
        assert got[-1] < got[0]
        # With clipping the two paths scale the gradients at different points (in place before Adam / inside the Adam
        # kernel): gradients that are pure rounding noise (the K third of qkv_bias: softmax is shift-invariant) then
        # normalise to +-lr with different signs — bounded by steps * lr, not a parity defect.
        rel, floor = (2e-3, 2e-4) if clip_norm == 0.0 else (2e-2, 2.5 * steps * lr)
        for (k, p), q in zip(model.named_parameters(), ref_model.parameters()):
            assert_close(p, q, rel, f"weights after {steps} steps: {k}", abs_floor=floor)
    else:
        # update every 2nd batch: batches 1 and 2 see the same weights, batch 3 sees the first update
        assert got[0] == got[1] and got[2] != got[1] and got[2] == got[3]
        assert opt.fused.step_count == steps // grad_accumulate
    cb.reducer.close()
