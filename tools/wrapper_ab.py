"""Host cost of the ops.gemm wrapper: the committed version against the working tree, same process.

    git show HEAD:carefree-learn_amd/ops.py > tools/_ops_prev.py    (git-ignored; travels with the gpurun snapshot)
    gpurun -- python tools/wrapper_ab.py
Round 3: 7.90 -> 6.53 us per call after reading shapes / strides once and inlining the checks."""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cflearn_amd  # noqa: E402
from cflearn_amd import ops as new_ops  # noqa: E402

spec = importlib.util.spec_from_file_location("cflearn_amd._ops_prev", os.path.join(ROOT, "tools", "_ops_prev.py"))
old_ops = importlib.util.module_from_spec(spec)
old_ops.__package__ = "cflearn_amd"
spec.loader.exec_module(old_ops)
dev = torch.device("cuda")
a = torch.randn(256, 128, device=dev).bfloat16()
b = torch.randn(128, 128, device=dev).bfloat16()
out = torch.empty(256, 128, dtype=torch.bfloat16, device=dev)
bias = torch.randn(128, device=dev)
for name, mod in (("previous", old_ops), ("working tree", new_ops), ("previous", old_ops), ("working tree", new_ops)):
    for _ in range(200):
        mod.gemm(a, b, bias=bias, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20000
    for _ in range(n):
        mod.gemm(a, b, bias=bias, out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name:14s} {1e6 * (t1 - t0) / n:6.2f} us of host time per ops.gemm call (tiny GEMM, the device keeps up: {1e6 * (time.perf_counter() - t0) / n:.2f} us incl. drain)")
