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
    torch.save(dict(g_sync=g_sync, g_acc=g_acc, p=arena.flat_p.clone(), offsets=arena.offsets,
                    extra_grad=extra.grad.clone()), f"{out}.{rank}")
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
