"""Is the GPU idle between the patch embedding and the first block of the stack in a NON-profiled run?  (rocprofv3 timelines show a
~0.5 ms hole there.)  HIP events on the issuing stream at: step start, MixingStackFn.forward entry, first block's first launch,
stack end, loss, end of backward, end of step.  python tools/gap_probe.py [batch]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cflearn_amd as C  # noqa: E402
from cflearn_amd import fused, ops  # noqa: E402
from cflearn_amd.engine import TrainStep  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model = C.vit_b16_classifier(1000).to(dev)
ts = TrainStep(model, lr=1e-4)
g = torch.Generator().manual_seed(1234)
img = torch.randn(BATCH, 3, 224, 224, generator=g).to(dev)
labels = torch.randint(0, 1000, (BATCH,), generator=g).to(dev)

marks = {}


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.setdefault(name, []).append(e)


orig_fwd = fused.MixingStackFn.forward
orig_block = fused._block_fwd
orig_ln = ops.layernorm_fwd
state = {"first_ln": False}


def fwd(ctx, x, *a):
    mark("stack_enter")
    state["first_ln"] = True
    out = orig_fwd(ctx, x, *a)
    mark("stack_exit")
    return out


def ln(*a, **k):
    if state["first_ln"]:
        state["first_ln"] = False
        mark("first_ln")
    return orig_ln(*a, **k)


fused.MixingStackFn.forward = staticmethod(fwd)
ops.layernorm_fwd = ln
orig_xent = ops.softmax_xent


def xent(*a, **k):
    mark("loss")
    return orig_xent(*a, **k)


ops.softmax_xent = xent
orig_launch = ts.optimizer.launch_step


def launch():
    mark("bwd_done")
    orig_launch()


ts.optimizer.launch_step = launch
N = 30
for i in range(N):
    mark("start")
    ts.step(img, labels)
mark("start")
torch.cuda.synchronize()
order = ["start", "stack_enter", "first_ln", "stack_exit", "loss", "bwd_done"]
for a, b in zip(order, order[1:]):
    d = [marks[a][i].elapsed_time(marks[b][i]) * 1e3 for i in range(N - 10, N)]
    print(f"{a:12s} -> {b:12s} median {statistics.median(d):9.1f} us  min {min(d):9.1f}")
d = [marks["bwd_done"][i].elapsed_time(marks["start"][i + 1]) * 1e3 for i in range(N - 10, N)]
print(f"{'bwd_done':12s} -> {'next start':12s} median {statistics.median(d):9.1f} us  min {min(d):9.1f}")
d = [marks["start"][i].elapsed_time(marks["start"][i + 1]) * 1e3 for i in range(N - 10, N)]
print(f"step median {statistics.median(d):9.1f} us")
