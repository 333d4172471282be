"""The RCCL calls of the distributed path on ONE GPU (world_size 1 is the only RCCL configuration a 1-GPU box allows):
process group on the `nccl` backend, parameter broadcast, bucketed all-reduce overlapped with the backward, the
autograd-aware embedding all-gather / reduce-scatter.  Runs in a child process so that the process group does not
outlive the test."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "oracle"))
    import torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import cflearn_amd as C
    from cflearn_amd.engine import TrainStep
    from cflearn_amd.contrastive import gather_rows_with_grad

    g = torch.load(os.path.join({root!r}, "tests", "golden", "vit_small.pt"))
    def build():
        cfg = dict(g["cfg"])
        m = C.build_module("cv_clf", config=dict(in_channels=3, num_classes=g["num_classes"], img_size=cfg.pop("img_size"),
                                                 latent_dim=cfg["latent_dim"], encoder="vit", encoder_config=cfg))
        m.load_state_dict(g["sd"])
        return m.to(dev)
    x, y = g["img"].to(dev), g["labels"].view(-1).to(dev)
    out = []
    for distributed in (False, True):
        ts = TrainStep(build(), lr=1e-3, weight_decay=0.01, distributed=distributed, bucket_bytes=1 << 16)
        if distributed:
            assert len(ts.reducer.buckets) >= 2
        for _ in range(3):
            ts.step(x, y)
        torch.cuda.synchronize()
        out.append(ts.arena.flat_p.clone())
    err = ((out[0] - out[1]).norm() / out[0].norm()).item()
    assert err < 2e-6, err   # W = 1: the exchange is the identity (LayerNorm parameter gradients: LDS float atomics)
    # embedding all-gather with gradient: ncclAllGather forward, ncclReduceScatter backward
    e = torch.randn(5, 8, device=dev, requires_grad=True)
    a = gather_rows_with_grad(e)
    assert torch.equal(a, e)
    w = torch.randn(5, 8, device=dev)
    (a * w).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(e.grad, w)
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL-1RANK-OK", err)
""")


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_rccl_code_path_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, port=_free_port())], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0 and "RCCL-1RANK-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
