"""LayerNorm backward at the ViT-B/16 step's shape (25216 x 768, f32 residual stream in, bf16 dy / dx / dx_add):
the round-1 launches (dx only / dgamma,dbeta only / both in the one-wave-per-row kernel) against the one-launch
half-wave-per-row kernel.  Algorithmic bytes: dy 2 + x 4 + dx_add 2 + dx 2 = 10 B / element."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cflearn_amd import ops

dev = torch.device("cuda")


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


for m, d in ((25216, 768), (12608, 768), (25216, 512), (16384, 1280)):
    x = torch.randn(m, d, device=dev)
    dy = torch.randn(m, d, device=dev).to(torch.bfloat16)
    add = torch.randn(m, d, device=dev).to(torch.bfloat16)
    w, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    _, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-6)
    pg = torch.zeros(2 * d, device=dev)
    nbytes = 10.0 * m * d
    t = timeit(lambda: ops.layernorm_fwd(x, w, b, 1e-6))
    print(f"{m}x{d} fwd (f32 in, bf16 out)            : {t * 1e6:7.1f} us  {6.0 * m * d / t / 1e12:5.2f} TB/s")
    ops.set_option("ln_bwd_fused", 0)
    t = timeit(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dx_add=add, want_param_grads=False))
    print(f"{m}x{d} r1 dx only                        : {t * 1e6:7.1f} us  {nbytes / t / 1e12:5.2f} TB/s")
    t = timeit(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dgamma=pg[:d], dbeta=pg[d:], want_dx=False))
    print(f"{m}x{d} r1 dgamma/dbeta only              : {t * 1e6:7.1f} us  {6.0 * m * d / t / 1e12:5.2f} TB/s")
    t = timeit(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dx_add=add, dgamma=pg[:d], dbeta=pg[d:]))
    print(f"{m}x{d} r1 both, one-wave-per-row kernel   : {t * 1e6:7.1f} us  {nbytes / t / 1e12:5.2f} TB/s")
    ops.set_option("ln_bwd_fused", 1)
    t = timeit(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dx_add=add, dgamma=pg[:d], dbeta=pg[d:]))
    print(f"{m}x{d} ONE launch, half-wave-per-row      : {t * 1e6:7.1f} us  {nbytes / t / 1e12:5.2f} TB/s (incl. the column reduce)")
