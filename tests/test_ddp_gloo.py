"""N > 1 path on CPU: two gloo ranks, arena-backed bucketed all-reduce.

DDP-mean semantics (the reference's intent): averaged per-rank grads == single-process
grads on the concatenated batch (fp32, summation-order tolerance 1e-6)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                               torch.nn.Linear(16, 3))


def _worker(rank: int, world: int, port: int, out: str) -> None:
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cflearn_amd as C

    assert C.get_ddp_info() == dict(rank=rank, world_size=world, local_rank=rank)
    model = _model()
    extra = torch.nn.Parameter(torch.ones(5))  # never used in forward: must not dead-lock a bucket
    params = list(model.parameters()) + [extra]
    if rank == 1:  # ranks start different; broadcast must fix it
        with torch.no_grad():
            for p in params:
                p.add_(1.0)
    arena = C.ParamArena(params, with_shadow=False)
    red = C.BucketedAllReduce(arena, bucket_bytes=256)  # tiny buckets -> several of them
    assert len(red.buckets) >= 3
    red.broadcast_parameters(0)
    torch.manual_seed(100)
    x = torch.randn(8, 6)
    y = torch.randn(8, 3)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]

    # step 1: plain synchronised step
    arena.zero_grad()
    ((model(xs) - ys) ** 2).mean().backward()
    red.finish()
    g_sync = arena.flat_g.clone()

    # step 2: gradient accumulation — first micro-batch without sync, second with
    arena.zero_grad()
    with red.no_sync():
        ((model(xs[:2]) - ys[:2]) ** 2).sum().backward()
        red.finish()  # no-op inside no_sync
    ((model(xs[2:]) - ys[2:]) ** 2).sum().backward()
    red.finish()
    g_acc = arena.flat_g.clone()

    # step 3 (round 5): the SAME reducer re-cut into other buckets between passes (bench.py's bucket sweep of a live job), and the
    # wire dtype switched: the plain step must give the same sum again (bf16 wire: to bf16 rounding of the per-rank gradients)
    g_rebucket = []
    for bb in (64, 1 << 20):
        nbk = red.rebucket(bb)
        assert nbk == len(red.buckets) and (nbk == 1) == (bb == 1 << 20)
        arena.zero_grad()
        ((model(xs) - ys) ** 2).mean().backward()
        red.finish()
        g_rebucket.append(arena.flat_g.clone())
    red.set_wire_bf16(True)
    arena.zero_grad()
    ((model(xs) - ys) ** 2).mean().backward()
    red.finish()
    g_wire = arena.flat_g.clone()
    red.set_wire_bf16(False)
    arena.zero_grad()
    ((model(xs) - ys) ** 2).mean().backward()
    try:
        red.rebucket(128)
        refused = False
    except RuntimeError:
        refused = True  # a pass is open: buckets may hold launched collectives
    red.finish()
    torch.save(dict(g_sync=g_sync, g_acc=g_acc, p=arena.flat_p.clone(), offsets=arena.offsets,
                    extra_grad=extra.grad.clone(), g_rebucket=g_rebucket, g_wire=g_wire, refused=refused,
                    g_after=arena.flat_g.clone()), f"{out}.{rank}")
    red.close()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_allreduce_two_ranks(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    # identical on every rank (sum then / W applied where there is no fused optimizer)
    assert torch.equal(r0["g_sync"], r1["g_sync"]) and torch.equal(r0["p"], r1["p"])
    assert torch.equal(r0["g_acc"], r1["g_acc"])
    assert r0["extra_grad"].abs().sum() == 0
    for r in (r0, r1):
        for gr in r["g_rebucket"]:
            assert torch.equal(gr, r["g_sync"])  # other buckets, the same element-wise sums
        assert (r["g_wire"] - r["g_sync"]).abs().max() <= 1e-2 * r["g_sync"].abs().max()
        assert r["refused"] and torch.equal(r["g_after"], r["g_sync"])

    # single-process reference on the concatenated batch
    model = _model()
    torch.manual_seed(100)
    x = torch.randn(8, 6)
    y = torch.randn(8, 3)
    ((model(x) - y) ** 2).mean().backward()
    flat = torch.zeros_like(r0["g_sync"])
    for p, off in zip(model.parameters(), r0["offsets"]):
        flat[off:off + p.numel()] = p.grad.reshape(-1)
    assert (r0["g_sync"] - flat).abs().max() < 1e-6
    # accumulation: sum over the 4 local samples per rank, averaged over ranks = sum over 8 / 2
    model.zero_grad()
    ((model(x) - y) ** 2).sum().backward()
    flat2 = torch.zeros_like(flat)
    for p, off in zip(model.parameters(), r0["offsets"]):
        flat2[off:off + p.numel()] = p.grad.reshape(-1) / 2
    assert (r0["g_acc"] - flat2).abs().max() < 1e-5


class _DirectLinear(torch.autograd.Function):
    """CPU stand-in for the HIP backward's protocol: the parameter gradients are written straight
    into `.grad` (the arena slice), `grad_ready_callbacks` are run, and None is returned for them —
    autograd then ALSO runs the parameters' post-accumulate hooks (the echo the reducer must ignore)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x)
        ctx.params = (w, b)
        return x @ w.detach().t() + b.detach()

    @staticmethod
    def backward(ctx, dy):
        from cflearn_amd.functional import grad_ready_callbacks

        (x,) = ctx.saved_tensors
        w, b = ctx.params
        w.grad.add_(dy.t() @ x)
        b.grad.add_(dy.sum(0))
        for prm in (w, b):
            for cb in grad_ready_callbacks:
                cb(prm)
        return dy @ w.detach(), None, None


def _direct_worker(rank: int, world: int, port: int, out: str) -> None:
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cflearn_amd as C

    model = _model()
    lins = [m for m in model if isinstance(m, torch.nn.Linear)]
    params = list(model.parameters())
    arena = C.ParamArena(params, with_shadow=False)
    red = C.BucketedAllReduce(arena, bucket_bytes=256)
    red.broadcast_parameters(0)
    torch.manual_seed(100)
    x = torch.randn(8, 6)
    y = torch.randn(8, 3)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]

    def fwd(inp):
        h = torch.tanh(_DirectLinear.apply(inp, lins[0].weight, lins[0].bias))
        h = torch.tanh(lins[1](h))  # autograd-accumulated layer between two direct-write layers
        return _DirectLinear.apply(h, lins[2].weight, lins[2].bias)

    grads = []
    for _ in range(2):  # two steps: the per-step bookkeeping must reset
        arena.zero_grad()
        ((fwd(xs) - ys) ** 2).mean().backward()
        red.finish()
        grads.append(arena.flat_g.clone())
    torch.save(dict(g0=grads[0], g1=grads[1], offsets=arena.offsets), f"{out}.{rank}")
    red.close()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_direct_write_notifications_two_ranks(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "res")
    mp.spawn(_direct_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["g0"], r1["g0"]) and torch.equal(r0["g0"], r0["g1"])
    model = _model()
    torch.manual_seed(100)
    x = torch.randn(8, 6)
    y = torch.randn(8, 3)
    ((model(x) - y) ** 2).mean().backward()
    flat = torch.zeros_like(r0["g0"])
    for p, off in zip(model.parameters(), r0["offsets"]):
        flat[off:off + p.numel()] = p.grad.reshape(-1)
    assert (r0["g0"] - flat).abs().max() < 1e-6


def _gather_worker(rank: int, world: int, port: int, out: str) -> None:
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cflearn_amd.contrastive import gather_rows_with_grad
    from clip_oracle import contrastive_loss_local

    torch.manual_seed(11)
    b, d = 3, 8
    img = torch.nn.functional.normalize(torch.randn(world * b, d), dim=-1)
    txt = torch.nn.functional.normalize(torch.randn(world * b, d), dim=-1)
    ls = torch.tensor(1.3, requires_grad=True)
    il = img[rank * b:(rank + 1) * b].clone().requires_grad_(True)
    tl = txt[rank * b:(rank + 1) * b].clone().requires_grad_(True)
    all_i, all_t = gather_rows_with_grad(il), gather_rows_with_grad(tl)
    assert torch.equal(all_i.detach(), img) and torch.equal(all_t.detach(), txt)  # rank-major order
    loss = contrastive_loss_local(il, tl, all_i, all_t, ls, offset=rank * b)
    loss.backward()
    torch.save(dict(loss=loss.detach(), gi=il.grad, gt=tl.grad, gls=ls.grad), f"{out}.{rank}")
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_embedding_gather_with_grad_two_ranks(tmp_path):
    """The "local loss + autograd-aware all-gather" scheme of contrastive.py: per-rank gradients, averaged over the
    ranks like DDP does, equal the gradients of the global-batch loss computed in one process."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from clip_oracle import contrastive_loss_local

    world, port = 2, _free_port()
    out = str(tmp_path / "gather")
    mp.spawn(_gather_worker, args=(world, port, out), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    torch.manual_seed(11)
    b, d = 3, 8
    img = torch.nn.functional.normalize(torch.randn(world * b, d), dim=-1).requires_grad_(True)
    txt = torch.nn.functional.normalize(torch.randn(world * b, d), dim=-1).requires_grad_(True)
    ls = torch.tensor(1.3, requires_grad=True)
    glob = contrastive_loss_local(img, txt, img, txt, ls)
    glob.backward()
    assert abs(sum(r["loss"] for r in res).item() / world - glob.item()) < 1e-6
    for r in range(world):
        # d(global mean loss)/d(features of rank r) = (1/W) * what rank r holds after the gather's backward
        assert (res[r]["gi"] / world - img.grad[r * b:(r + 1) * b]).abs().max() < 1e-6
        assert (res[r]["gt"] / world - txt.grad[r * b:(r + 1) * b]).abs().max() < 1e-6
    assert abs(sum(r["gls"] for r in res).item() / world - ls.grad.item()) < 1e-6


def _callback_worker(rank: int, world: int, port: int, out: str) -> None:
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cflearn_amd.ddp import RcclDDPCallback

    model = _model()
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(3.0)  # the broadcast of before_loop must undo this

    class _Obj:
        pass

    trainer = _Obj()
    trainer.model = _Obj()
    trainer.model.m = model
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    trainer.optimizers = {"all": opt}
    cb = RcclDDPCallback(bucket_bytes=256)
    cb.before_loop(trainer)  # reference trainer.py:312-313
    torch.manual_seed(100)
    x, y = torch.randn(8, 6), torch.randn(8, 3)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for _ in range(3):  # the reference's step: zero_grad (set_to_none), backward, optimizer.step
        opt.zero_grad()
        ((model(xs) - ys) ** 2).mean().backward()
        opt.step()
    torch.save([p.detach().clone() for p in model.parameters()], f"{out}.{rank}")
    cb.reducer.close()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_trainer_callback_with_set_to_none_zero_grad(tmp_path):
    """`RcclDDPCallback` under the reference trainer's own step (torch optimizer, `zero_grad()` with set_to_none): three
    SGD steps on two ranks == three steps in one process on the concatenated batch."""
    world, port = 2, _free_port()
    out = str(tmp_path / "cb")
    mp.spawn(_callback_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)
    model = _model()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    torch.manual_seed(100)
    x, y = torch.randn(8, 6), torch.randn(8, 3)
    for _ in range(3):
        opt.zero_grad()
        ((model(x) - y) ** 2).mean().backward()
        opt.step()
    for a, p in zip(r0, model.parameters()):
        assert (a - p.detach()).abs().max() < 1e-6


def _callback_fused_worker(rank: int, world: int, port: int, out: str) -> None:
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cflearn_amd.ddp import RcclDDPCallback
    from cflearn_amd.optim import FusedAdamWOptimizer

    model = _model()
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)

    class _Obj:
        pass

    trainer = _Obj()
    trainer.model = _Obj()
    trainer.model.m = model
    opt = FusedAdamWOptimizer(model.parameters(), lr=1e-3)
    trainer.optimizers = {"all": opt}
    cb = RcclDDPCallback(bucket_bytes=256)
    cb.before_loop(trainer)
    assert cb.reducer.arena is opt.arena            # the optimizer's own arena is the one that is reduced
    assert opt.fused.grad_scale == 1.0              # the callback leaves the rank AVERAGE in the arena (clipping reads it)
    opt.zero_grad()
    torch.manual_seed(100)
    x, y = torch.randn(8, 6), torch.randn(8, 3)
    ((model(x[rank * 4:(rank + 1) * 4]) - y[rank * 4:(rank + 1) * 4]) ** 2).mean().backward()
    try:
        opt.step()  # pre-step hook = the gradient exchange; the update itself needs the GPU
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    torch.save(dict(p=opt.arena.flat_p.clone(), g=opt.arena.flat_g.clone()), f"{out}.{rank}")
    cb.reducer.close()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_trainer_callback_reuses_the_fused_optimizers_arena(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "cbf")
    mp.spawn(_callback_fused_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["p"], r1["p"])  # broadcast from rank 0
    assert torch.equal(r0["g"], r1["g"])  # the rank average, already in the arena when backward() returned
    model = _model()
    torch.manual_seed(100)
    x, y = torch.randn(8, 6), torch.randn(8, 3)
    ((model(x) - y) ** 2).mean().backward()
    flat = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 8)) for p in model.parameters()])
    assert (r0["g"] - flat).abs().max() < 1e-6


def _callback_clip_accum_worker(rank: int, world: int, port: int, out: str, clip: float, accumulate: int) -> None:
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cflearn_amd.ddp import RcclDDPCallback

    model = _model()

    class _Obj:
        pass

    trainer = _Obj()
    trainer.model = _Obj()
    trainer.model.m = model
    step_obj = _Obj()
    step_obj.grad_accumulate = None
    trainer.model.train_steps = [step_obj]
    trainer.state = _Obj()
    trainer.state.step = 0
    trainer.config = _Obj()
    trainer.config.grad_accumulate = accumulate
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    trainer.optimizers = {"all": opt}
    cb = RcclDDPCallback(bucket_bytes=256)
    cb.before_loop(trainer)
    torch.manual_seed(100)
    launches = []
    orig = cb.reducer._launch
    cb.reducer._launch = lambda b: (launches.append(trainer.state.step), orig(b))[1]
    for it in range(4):  # the reference's update (schema.py:977-986, 1277-1282): backward; if update: clip, step, zero
        trainer.state.step += 1
        x, y = torch.randn(8, 6), torch.randn(8, 3)
        ((model(x[rank * 4:(rank + 1) * 4]) - y[rank * 4:(rank + 1) * 4]) ** 2).mean().backward()
        if trainer.state.step % accumulate == 0:
            if clip > 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip)  # what accelerator.clip_grad_norm_ runs
            opt.step()
            opt.zero_grad()
    torch.save(dict(p=[p.detach().clone() for p in model.parameters()], launches=launches), f"{out}.{rank}")
    cb.reducer.close()
    dist.destroy_process_group()


def _single_process_clip_accum(clip: float, accumulate: int):
    model = _model()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    torch.manual_seed(100)
    for it in range(1, 5):
        x, y = torch.randn(8, 6), torch.randn(8, 3)
        ((model(x) - y) ** 2).mean().backward()
        if it % accumulate == 0:
            if clip > 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
            opt.step()
            opt.zero_grad()
    return [p.detach() for p in model.parameters()]


@pytest.mark.timeout(120)
@pytest.mark.parametrize("clip,accumulate", [(0.05, 1), (0.0, 2), (0.05, 2)])
def test_trainer_callback_clipping_and_accumulation(tmp_path, clip, accumulate):
    """with clip_norm > 0 the reference clips between backward and optimizer.step — the exchange must be
    complete and AVERAGED by then (end-of-backward finish); with grad_accumulate > 1 only the update pass reduces."""
    world, port = 2, _free_port()
    out = str(tmp_path / "cca")
    mp.spawn(_callback_clip_accum_worker, args=(world, port, out, clip, accumulate), nprocs=world, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for a, b in zip(r0["p"], r1["p"]):
        assert torch.equal(a, b)
    want = _single_process_clip_accum(clip, accumulate)
    for a, w in zip(r0["p"], want):
        assert (a - w).abs().max() < 1e-6
    # all-reduces were launched on update passes only
    assert r0["launches"] and all(s % accumulate == 0 for s in r0["launches"])


class _LossWithParam(torch.nn.Module):
    """a loss module that owns a parameter (the reference prepares `model.all_modules`, loss included: trainer.py:268-272)"""

    def __init__(self) -> None:
        super().__init__()
        self.log_scale = torch.nn.Parameter(torch.zeros(()))
        self.register_buffer("seen", torch.zeros((), dtype=torch.long))

    def forward(self, pred, y):
        return ((pred - y) ** 2).mean() * torch.exp(self.log_scale)


def _bn_model():
    torch.manual_seed(11)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.BatchNorm1d(16), torch.nn.Tanh(), torch.nn.Linear(16, 3))


def _callback_buffers_worker(rank: int, world: int, port: int, out: str) -> None:
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cflearn_amd.ddp import RcclDDPCallback

    model, loss = _bn_model(), _LossWithParam()
    if rank == 1:  # rank 1 starts from other weights AND other buffers: the callback must bring both in line with rank 0
        with torch.no_grad():
            for p in list(model.parameters()) + list(loss.parameters()):
                p.add_(1.0)
            model[1].running_mean.add_(3.0)
            model[1].running_var.mul_(5.0)
            model[1].num_batches_tracked.add_(7)
            loss.seen.add_(9)

    class _Obj:
        pass

    trainer = _Obj()
    trainer.model = _Obj()
    trainer.model.m, trainer.model.loss = model, loss
    trainer.model.all_modules = [model]  # the loss is found through `model.loss`
    opt = torch.optim.SGD(list(model.parameters()) + list(loss.parameters()), lr=0.05)
    trainer.optimizers = {"all": opt}
    cb = RcclDDPCallback(bucket_bytes=256)
    cb.before_loop(trainer)
    assert cb.reducer.comm is None  # gloo: the ProcessGroup launches the collectives
    assert any(p is loss.log_scale for p in cb.reducer.arena.params)
    buffers_after = [b.detach().clone() for b in list(model.buffers()) + list(loss.buffers())]
    torch.manual_seed(100)
    x, y = torch.randn(8, 6), torch.randn(8, 3)
    for _ in range(2):
        opt.zero_grad()
        loss(model(x[rank * 4:(rank + 1) * 4]), y[rank * 4:(rank + 1) * 4]).backward()
        opt.step()
    torch.save(dict(buffers=buffers_after, params=[p.detach().clone() for p in list(model.parameters()) + list(loss.parameters())],
                    n_buckets=len(cb.reducer.buckets), tail=len(cb.reducer.buckets[-1].param_ids)), f"{out}.{rank}")
    cb.reducer.close()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_trainer_callback_broadcasts_buffers_and_hooks_the_loss_parameters(tmp_path):
    """the DDP constructor the callback replaces also synchronised module BUFFERS (BatchNorm running
    statistics of the MNIST / FCNN configurations) and wrapped every prepared module, the loss included."""
    world, port = 2, _free_port()
    out = str(tmp_path / "cbb")
    mp.spawn(_callback_buffers_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    ref_model, ref_loss = _bn_model(), _LossWithParam()
    want = [b.detach().clone() for b in list(ref_model.buffers()) + list(ref_loss.buffers())]
    for a, b, w in zip(r0["buffers"], r1["buffers"], want):  # right after before_loop: rank 0's (= the seed's) values everywhere
        assert torch.equal(a, b) and torch.equal(a, w)
    for a, b in zip(r0["params"], r1["params"]):  # weights AND the loss parameter stay in lock-step
        assert torch.equal(a, b)
    assert (r0["params"][-1] - 0.0).abs() > 0  # the loss parameter was trained (its gradient was exchanged and applied)


def _tail_bucket_worker(rank: int, world: int, port: int, out: str) -> None:
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cflearn_amd as C
    from cflearn_amd import functional as HF

    torch.manual_seed(3)
    layers = [torch.nn.Linear(8, 8) for _ in range(6)]
    model = torch.nn.Sequential(*layers)
    arena = C.ParamArena(list(model.parameters()), with_shadow=False)
    red = C.BucketedAllReduce(arena, bucket_bytes=2 * 72 * 4, tail_bytes=72 * 4, finish_after_backward=True, average=True)
    # the LAST bucket (first-registered parameters) holds just the stem: one Linear (weight 64 + bias 8 floats)
    sizes = [sum(arena.params[i].numel() for i in b.param_ids) for b in red.buckets]
    launched_at_end = []
    # a checkpointed LAST block: its inner backward is a nested graph task; the end-of-backward callback must still
    # belong to the whole pass: all buckets reduced exactly once, gradients == the rank average
    torch.manual_seed(50 + rank)
    x = torch.randn(4, 8)

    class DirectLinear(torch.autograd.Function):
        """what the HIP Functions do: the parameter gradients are written straight into `.grad` (the arena) inside
        backward, announced through `grad_ready_callbacks`, and None is returned for them"""

        @staticmethod
        def forward(ctx, h, w, b):
            ctx.save_for_backward(h, w)
            ctx.prm = (w, b)
            return h @ w.t() + b

        @staticmethod
        def backward(ctx, dy):
            h, w = ctx.saved_tensors
            wp, bp = ctx.prm
            with torch.no_grad():
                wp.grad.add_(dy.t() @ h)
                bp.grad.add_(dy.sum(0))
            for prm in (wp, bp):
                for cb in HF.grad_ready_callbacks:
                    cb(prm)
            return dy @ w, None, None

    def tail_block(h):
        return DirectLinear.apply(torch.tanh(layers[4](h)), layers[5].weight, layers[5].bias)

    arena.zero_grad()
    h = x
    for lyr in layers[:4]:
        h = torch.tanh(lyr(h))
    out_t = HF.gradient_checkpoint(tail_block, (h,), list(layers[4].parameters()) + list(layers[5].parameters()), True)
    calls = {"n": 0}
    orig = red._launch

    def counting(b):
        calls["n"] += 1
        return orig(b)

    red._launch = counting
    out_t.pow(2).mean().backward()
    launched_at_end.append(calls["n"])
    g = arena.flat_g.clone()
    # reference: plain backward of the same graph, local gradient
    arena.zero_grad()
    red.sync_enabled = False
    h = x
    for lyr in layers[:4]:
        h = torch.tanh(lyr(h))
    tail_block(h).pow(2).mean().backward()
    local = arena.flat_g.clone()
    red.sync_enabled = True
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    torch.save(dict(sizes=sizes, g=g, avg=sum(both) / world, launches=launched_at_end, n_buckets=len(red.buckets)), f"{out}.{rank}")
    red.close()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_small_tail_bucket_and_checkpointed_last_block(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "tail")
    mp.spawn(_tail_bucket_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["sizes"][-1] == 72 and sum(r0["sizes"]) == 6 * 72 and r0["sizes"][0] == 144  # tail = the stem only
    assert r0["launches"] == [r0["n_buckets"]]  # every bucket launched exactly once in the checkpointed pass
    assert torch.equal(r0["g"], r1["g"])
    assert (r0["g"] - r0["avg"]).abs().max() < 1e-6


def _deferred_flush_worker(rank: int, world: int, port: int, out: str) -> None:
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cflearn_amd as C
    from cflearn_amd import functional as HF

    torch.manual_seed(3)
    layers = [torch.nn.Linear(8, 8) for _ in range(4)]
    arena = C.ParamArena([p for lyr in layers for p in lyr.parameters()], with_shadow=False)
    red = C.BucketedAllReduce(arena, bucket_bytes=72 * 4, tail_bytes=72 * 4, finish_after_backward=True, average=True)
    queue: list = []

    def flush() -> None:  # what fused._flush_deferred does with its queued weight gradients
        while queue:
            (wp, bp), dy, h = queue.pop(0)
            with torch.no_grad():
                wp.grad.add_(dy.t() @ h)
                bp.grad.add_(dy.sum(0))
            for prm in (wp, bp):
                for cb in HF.grad_ready_callbacks:
                    cb(prm)

    class DeferredLinear(torch.autograd.Function):
        """parameter gradients QUEUED in backward (one grouped launch later).  The parameters are NOT tensor inputs of the
        Function (as in the fused block stack): autograd then has no AccumulateGrad edge whose post-accumulate hook would
        announce the parameter — with nothing written yet — the moment this backward returns."""

        @staticmethod
        def forward(ctx, h, lyr):
            w = lyr.weight.detach()
            ctx.save_for_backward(h, w)
            ctx.prm = (lyr.weight, lyr.bias)
            return h @ w.t() + lyr.bias.detach()

        @staticmethod
        def backward(ctx, dy):
            h, w = ctx.saved_tensors
            queue.append((ctx.prm, dy, h))
            return dy @ w, None

    def net(x, deferred):
        h = x
        for i, lyr in enumerate(layers):
            # the FIRST layer's gradients are the last of the pass: still queued when autograd's end-of-backward callbacks run
            h = DeferredLinear.apply(h, lyr) if (deferred and i in (0, 2)) else lyr(h)
            h = torch.tanh(h)
        return h.pow(2).mean()

    torch.manual_seed(60 + rank)
    x = torch.randn(5, 8, requires_grad=True)  # (layer 0 is deferred: its Function needs an input that wants a gradient)
    HF.deferred_grad_flushes.append(flush)
    try:
        arena.zero_grad()
        net(x, True).backward()
        g = arena.flat_g.clone()
        leftover = len(queue)
    finally:
        HF.deferred_grad_flushes.remove(flush)
    arena.zero_grad()
    red.sync_enabled = False
    net(x, False).backward()
    local = arena.flat_g.clone()
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    torch.save(dict(g=g, avg=sum(both) / world, leftover=leftover), f"{out}.{rank}")
    red.close()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gradients_queued_for_a_later_launch_are_flushed_before_the_pass_is_finished(tmp_path):
    """Round 3: weight gradients may sit in a queue (fused.queue_linear_dw) when the backward pass ends; the reducer's
    end-of-backward callback runs the registered flushes (functional.deferred_grad_flushes) BEFORE it finishes the pass, whatever
    the order autograd runs its callbacks in: every bucket reduced, gradients = the rank average."""
    world, port = 2, _free_port()
    out = str(tmp_path / "defer")
    mp.spawn(_deferred_flush_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["leftover"] == 0 and r1["leftover"] == 0
    assert torch.equal(r0["g"], r1["g"])
    assert (r0["g"] - r0["avg"]).abs().max() < 1e-6
