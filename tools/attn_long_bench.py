"""General-length attention kernels at the DDPM UNet's shapes (B x 8 heads, T = (img / rate)^2 tokens, head_dim 40 / 80 / 160)
and a long-sequence head_dim-64 case: forward, dQ pass, dK/dV pass, event-timed, FLOPs counted at the true head_dim.

    python tools/attn_long_bench.py [--quick]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


cases = [  # (B, H, T, dh, reps)
    (8, 8, 4096, 40, 10), (8, 8, 1024, 80, 10), (8, 8, 256, 160, 10),      # UNet 64^2 x 8
    (1, 8, 65536, 40, 2), (1, 8, 16384, 80, 4), (1, 8, 4096, 160, 10),     # UNet 256^2 x 1
    (4, 12, 4096, 64, 5),
]
for a in sys.argv[1:]:
    if a.startswith("--two-tiles="):
        ops.set_option("attn_two_tiles", int(a.split("=")[1]))
if "--quick" in sys.argv:
    cases = [c for c in cases if c[2] <= 16384]
g = torch.Generator(device=dev).manual_seed(0)
_w = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
for _ in range(200):  # ~0.2 s of matmuls: the first timed case of a cold process is 20 % slow otherwise (clocks ramping)
    _w @ _w
torch.cuda.synchronize()
for b, h, t, dh, reps in cases:
    d = h * dh
    rnd = lambda: (torch.randn(b, t, d, device=dev, generator=g) * 0.5).to(torch.bfloat16)  # noqa: E731
    q, k, v, d_o = rnd(), rnd(), rnd(), rnd()
    o, lse = ops.attn_fwd(q, k, v, h, head_dim=dh)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(b, h, t, device=dev)
    fl = 4.0 * b * h * t * t * dh
    f = timeit(lambda: ops.attn_fwd(q, k, v, h, head_dim=dh), reps)
    t1 = timeit(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, h, dq=dq, dk=dk, dv=dv, parts=1, delta=delta, head_dim=dh), reps)
    t2 = timeit(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, h, dq=dq, dk=dk, dv=dv, parts=2, delta=delta, head_dim=dh), reps)
    t3 = timeit(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, h, dq=dq, dk=dk, dv=dv, parts=3, delta=delta, head_dim=dh), reps)
    print(f"B{b} H{h} T{t:6d} dh{dh:4d} | fwd {f:9.1f} us {fl / f / 1e6:5.0f} TF | dq {t1:9.1f} us {1.5 * fl / t1 / 1e6:5.0f} TF | "
          f"dkv {t2:9.1f} us {2.0 * fl / t2 / 1e6:5.0f} TF | bwd (one call) {t3:9.1f} us {2.5 * fl / t3 / 1e6:5.0f} TF (5 products counted)", flush=True)
