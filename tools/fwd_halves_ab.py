"""Whole-step A/B of the forward run as 1 / 2 / 3 batch-slice pipelines on separate streams (fused.FWD_HALVES), after checking
that the logits and gradients are bit-equal.   python tools/fwd_halves_ab.py [batch]"""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cflearn_amd as C
from cflearn_amd import fused
from cflearn_amd.engine import TrainStep

BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 128
VALUES = [int(v) for v in os.environ.get("HALVES", "1,2,3,2,1").split(",")]
dev = torch.device("cuda")
torch.manual_seed(0)
model = C.vit_b16_classifier(1000).to(dev)
g = torch.Generator().manual_seed(1234)
img = torch.randn(BATCH, 3, 224, 224, generator=g).to(dev)
labels = torch.randint(0, 1000, (BATCH,), generator=g).to(dev)

outs = {}
for v in (1, 2, 3):
    fused.FWD_HALVES = v
    model.zero_grad(set_to_none=True)
    y = model(img)
    y = y["predictions"] if isinstance(y, dict) else y
    y.float().square().mean().backward()
    torch.cuda.synchronize()
    outs[v] = (y.detach().clone(), [p.grad.detach().clone() for p in model.parameters()])
for v in (2, 3):
    same = torch.equal(outs[1][0], outs[v][0]) and all(torch.equal(a, b) for a, b in zip(outs[1][1], outs[v][1]))
    print(f"FWD_HALVES {v}: logits and all gradients bit-equal to the one-stream forward: {same}", flush=True)
model.zero_grad(set_to_none=True)
ts = TrainStep(model, lr=1e-4, use_graph=False)


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step(img, labels)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run(5)
res = {}
for rnd in range(4):
    for v in VALUES:
        fused.FWD_HALVES = v
        res.setdefault(v, []).append(run(10))
for v, r in sorted(res.items()):
    print(f"step, forward in {v} slice(s): median {statistics.median(r):7.3f} ms  min {min(r):7.3f}  n={len(r)}")
