"""Host and device time of one ViT-B/16 step with and without launch plans (fused.StackPlan): where does a replayed step spend
its host time?  python tools/plan_probe.py [batch]"""
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


def run(plans: bool):
    fused.STACK_PLANS = plans
    fused._plans.clear()
    torch.manual_seed(0)
    model = C.vit_b16_classifier(1000).to(dev)
    ts = TrainStep(model, lr=1e-4)
    t_replay = []
    orig = fused._replay

    def timed(ops_):
        t0 = time.perf_counter()
        orig(ops_)
        t_replay.append((time.perf_counter() - t0, len(ops_)))

    fused._replay = timed
    for _ in range(6):
        ts.step(img, labels)
    torch.cuda.synchronize()
    t_replay.clear()
    host, dev_ms = [], []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        t0 = time.perf_counter()
        ts.step(img, labels)
        host.append((time.perf_counter() - t0) * 1e3)
        e1.record()
        torch.cuda.synchronize()
        dev_ms.append(e0.elapsed_time(e1))
    fused._replay = orig
    print(f"plans={plans}: host issue {statistics.median(host):.2f} ms, device (issue-limited, sync per step) {statistics.median(dev_ms):.2f} ms")
    if t_replay:
        per = {}
        for dt, n in t_replay:
            per.setdefault(n, []).append(dt)
        for n, ds in per.items():
            print(f"    replay of {n} entries: {statistics.median(ds) * 1e3:.2f} ms ({statistics.median(ds) / n * 1e6:.2f} us / entry)")
    plan = next(iter(fused._plans.values())) if fused._plans else None
    if plan is not None:
        kinds = [k for k, _, _ in plan.fwd + plan.bwd]
        print(f"    plan: fwd {len(plan.fwd)} bwd {len(plan.bwd)} entries; launches {kinds.count(0)}, stream ops {kinds.count(1)}, notifications {kinds.count(2)}; kept tensors {len(plan.keep)}")
    # back-to-back throughput
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ts.step(img, labels)
    torch.cuda.synchronize()
    print(f"    20 steps back to back: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms / step")


run(False)
run(True)
