"""Where the two CLIP towers' forward and backward passes sit in an UNTRACED step: HIP events on the stream each block stack's Function
runs on (fused.MixingStackFn forward / backward: one call per tower and pass), relative to the step's first launch.
    python tools/clip_phase_probe.py [steps]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from cflearn_amd import functional as HF  # noqa: E402
from cflearn_amd import fused  # noqa: E402

marks = []


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record(HF.cur_stream())
    return e


def wrap(name, fn):
    def inner(ctx, *a, **kw):
        rows = None
        for t in a:
            if isinstance(t, torch.Tensor) and t.dim() == 3:
                rows = t.shape[0] * t.shape[1]
                break
        e0 = ev()
        out = fn(ctx, *a, **kw)
        marks.append((f"{name} rows={rows}", e0, ev()))
        return out
    return staticmethod(inner)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    args = argparse.Namespace(workload="clip", batch=128, img=64, graph=False, steps=1, warmup=3, gemm_table=False, no_step_in_backward=False)
    step, *_ = bench.build_other_workload(args)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    f0, b0 = fused.MixingStackFn.forward, fused.MixingStackFn.backward
    fused.MixingStackFn.forward = wrap("fwd", f0)
    fused.MixingStackFn.backward = wrap("bwd", b0)
    for i in range(steps):
        marks.clear()
        s0 = torch.cuda.Event(enable_timing=True)
        s0.record(HF.cur_stream())
        step()
        s1 = torch.cuda.Event(enable_timing=True)
        s1.record(HF.cur_stream())
        torch.cuda.synchronize()
        print(f"step {i}: {s0.elapsed_time(s1):.2f} ms on the issuing stream")
        for name, e0, e1 in marks:
            print(f"   {name:<22} start +{s0.elapsed_time(e0):6.2f} ms   end +{s0.elapsed_time(e1):6.2f} ms   ({e0.elapsed_time(e1):5.2f} ms)")


if __name__ == "__main__":
    main()
