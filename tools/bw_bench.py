"""HBM bandwidth calibration: torch fill / copy vs our element-wise kernels, and LayerNorm kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cflearn_amd import ops

dev = torch.device("cuda")

def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps

for mb in (64, 256, 1024):
    n = mb * (1 << 20) // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    y = torch.empty_like(x)
    xf = torch.empty(n // 2, dtype=torch.float32, device=dev).normal_()
    t = timeit(lambda: y.fill_(1.0)); print(f"{mb:5d} MB  torch fill      : {mb / 1024 / t / 1.024:7.2f} TB/s written")
    t = timeit(lambda: y.copy_(x)); print(f"{mb:5d} MB  torch copy      : {2 * mb / 1024 / t / 1.024:7.2f} TB/s (r+w)")
    t = timeit(lambda: ops.add(x, x)); print(f"{mb:5d} MB  cfhip add (2r1w): {3 * mb / 1024 / t / 1.024:7.2f} TB/s")
    t = timeit(lambda: ops.to_bf16(xf)); print(f"{mb:5d} MB  cfhip f32->bf16 : {1.5 * mb / 1024 / t / 1.024:7.2f} TB/s")
    t = timeit(lambda: ops.gelu_fwd(x)); print(f"{mb:5d} MB  cfhip gelu      : {2 * mb / 1024 / t / 1.024:7.2f} TB/s")
m, d = 12608, 768
x = torch.randn(m, d, device=dev).to(torch.bfloat16); w = torch.ones(d, device=dev); b = torch.zeros(d, device=dev)
t = timeit(lambda: ops.layernorm_fwd(x, w, b, 1e-6)); print(f"LN fwd {m}x{d}: {t*1e6:.1f} us = {2*m*d*2/t/1e12:.2f} TB/s")
y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-6)
t = timeit(lambda: ops.layernorm_bwd(x, x, w, mean, rstd, dx_add=x)); print(f"LN bwd(+add) {m}x{d}: {t*1e6:.1f} us = {4*m*d*2/t/1e12:.2f} TB/s")
xx = torch.randn(m, 3072, device=dev).to(torch.bfloat16)
t = timeit(lambda: ops.colsum(xx)); print(f"colsum {m}x3072: {t*1e6:.1f} us = {m*3072*2/t/1e12:.2f} TB/s")
t = timeit(lambda: ops.colsum(x)); print(f"colsum {m}x768: {t*1e6:.1f} us = {m*768*2/t/1e12:.2f} TB/s")
