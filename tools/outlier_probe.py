"""How often does a training step take much longer than the median?  300 steps per variant, HIP events between steps.
python tools/outlier_probe.py"""
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
g = torch.Generator().manual_seed(1234)
img = torch.randn(128, 3, 224, 224, generator=g).to(dev)
labels = torch.randint(0, 1000, (128,), generator=g).to(dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for plans, sib in ((False, False), (False, True), (True, False), (True, True)):
    fused.STACK_PLANS = plans
    fused._plans.clear()
    torch.manual_seed(0)
    ts = TrainStep(C.vit_b16_classifier(1000).to(dev), lr=1e-4, step_in_backward=sib)
    for _ in range(8):
        ts.step(img, labels)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    host = []
    marks[0].record()
    for i in range(N):
        t0 = time.perf_counter()
        ts.step(img, labels)
        host.append((time.perf_counter() - t0) * 1e3)
        marks[i + 1].record()
    torch.cuda.synchronize()
    d = [marks[i].elapsed_time(marks[i + 1]) for i in range(N)]
    slow = [(i, round(x, 1), round(host[i], 1)) for i, x in enumerate(d) if x > 1.3 * statistics.median(d)]
    print(f"plans={plans} step_in_backward={sib}: median {statistics.median(d):.3f} ms, mean {statistics.mean(d):.3f}, max {max(d):.1f}; "
          f"host median {statistics.median(host):.2f} max {max(host):.1f}; slow steps (index, device ms, host ms): {slow[:8]}", flush=True)
    del ts
    torch.cuda.empty_cache()
