"""Which ATen operators (the ones that launch a kernel or a copy) does one training step still issue, and from where?
A TorchDispatchMode logs every aten op of one step of a workload with the innermost frame inside this package.
    python tools/aten_trace.py unet|clip|vit"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402

VIEW_OPS = ("view", "reshape", "permute", "transpose", "slice", "select", "expand", "unsqueeze", "squeeze", "as_strided", "detach", "alias",
            "t.default", "unbind", "split", "narrow", "_unsafe_view", "empty", "size", "stride", "is_", "sym_", "lift_fresh", "_local_scalar")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW_OPS):
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=24)):
                if "carefree-learn_amd" in fr.filename or fr.filename.endswith("bench.py"):
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            shape = ""
            for a in args:
                if isinstance(a, torch.Tensor):
                    shape = f"{tuple(a.shape)} {str(a.dtype).replace('torch.', '')}"
                    break
            self.counts[(name, where, shape)] += 1
        return func(*args, **(kwargs or {}))


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "unet"
    import argparse

    args = argparse.Namespace(workload=wl, batch=128, img=64, graph=False, steps=1, warmup=3, gemm_table=False, no_step_in_backward=False)
    step, *_ = bench.build_other_workload(args) if hasattr(bench, "build_other_workload") else (None,)
    if step is None:
        raise SystemExit("bench.build_other_workload missing")
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    log = Log()
    with log:
        step()
    torch.cuda.synchronize()
    by_op = collections.Counter()
    for (name, where, shape), n in log.counts.items():
        by_op[name] += n
    print("== aten ops of one step (views and allocations not listed)")
    for name, n in by_op.most_common(25):
        print(f"{n:6d}  {name}")
    print("== by call site")
    for (name, where, shape), n in log.counts.most_common(60):
        print(f"{n:6d}  {name:34s} {where:44s} {shape}")


if __name__ == "__main__":
    main()
