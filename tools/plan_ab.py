"""Interleaved A/B of the whole ViT-B/16 step with launch plans on / off in ONE process (two engines on two models).
python tools/plan_ab.py [batch]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cflearn_amd as C  # noqa: E402
from cflearn_amd import fused  # noqa: E402
from cflearn_amd.engine import TrainStep  # noqa: E402

dev = torch.device("cuda")
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.Generator().manual_seed(1234)
img = torch.randn(BATCH, 3, 224, 224, generator=g).to(dev)
labels = torch.randint(0, 1000, (BATCH,), generator=g).to(dev)
eng = {}
for name, plans in (("plans off", False), ("plans on", True)):
    torch.manual_seed(0)
    eng[name] = (TrainStep(C.vit_b16_classifier(1000).to(dev), lr=1e-4), plans)


def run(name, n):
    ts, plans = eng[name]
    fused.STACK_PLANS = plans
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step(img, labels)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3


for name in eng:
    run(name, 5)
res = {k: [] for k in eng}
for rnd in range(6):
    for name in eng:
        res[name].append(run(name, 10))
for k, v in res.items():
    print(f"{k:10s} step median {statistics.median(x[0] for x in v):7.3f} ms  min {min(x[0] for x in v):7.3f}   host issue median {statistics.median(x[1] for x in v):6.2f} ms")
