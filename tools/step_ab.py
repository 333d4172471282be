"""Within-process A/B of whole-step variants (interleaved rounds, medians): python tools/step_ab.py"""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cflearn_amd as C
from cflearn_amd import ops
from cflearn_amd.engine import TrainStep
from cflearn_amd.functional import SideStream

dev = torch.device("cuda")
torch.manual_seed(0)
model = C.vit_b16_classifier(1000).to(dev)
ts = TrainStep(model, lr=1e-4, use_graph=False)
g = torch.Generator().manual_seed(1234)
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 64
img = torch.randn(BATCH, 3, 224, 224, generator=g).to(dev)
labels = torch.randint(0, 1000, (BATCH,), generator=g).to(dev)

HIGH = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[1] if False else -1)
VARIANTS = {
    "shipped (main = default stream)": dict(high=False),
    "whole step issued on a HIGH-priority stream (side streams stay normal)": dict(high=True),
}
STATE = dict(high=False)

def apply(v):
    STATE["high"] = v["high"]


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if STATE["high"]:
        HIGH.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(HIGH):
            for _ in range(n):
                ts.step(img, labels)
    else:
        for _ in range(n):
            ts.step(img, labels)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for v in VARIANTS.values():
    apply(v); run(3)
res = {k: [] for k in VARIANTS}
for rnd in range(5):
    for k, v in VARIANTS.items():
        apply(v)
        res[k].append(run(10))
for k, v in res.items():
    print(f"{k:68s} median {statistics.median(v):7.3f} ms  min {min(v):7.3f}  all {[round(x, 2) for x in v]}")
