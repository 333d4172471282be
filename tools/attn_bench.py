"""Attention kernels on the ViT-B/16 shape (packed qkv, T = 197, 12 heads of 64): python tools/attn_bench.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cflearn_amd as C
from cflearn_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H, T, D = 12, 197, 768
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B, T, 3 * D, device=dev, generator=g) * 0.5).bfloat16()
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
d_o = (torch.randn(B, T, D, device=dev, generator=g) * 0.1).bfloat16()
dqkv = torch.empty_like(qkv)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


flop_f = 4.0 * B * H * T * T * 64
io_f = (3 + 1) * B * T * D * 2
for ab in [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]:
    ops.set_option("attn_ablate", ab)
    us = timeit(lambda: ops.attn_fwd(q, k, v, H))
    print(f"fwd  ablate={ab}: {us:7.1f} us  {flop_f / us / 1e6:6.0f} TFLOP/s  {io_f / us / 1e6:5.2f} TB/s (algorithmic)")
ops.set_option("attn_ablate", 0)
o, lse = ops.attn_fwd(q, k, v, H)
for ab in (1, 2, 4, 5):
    ops.set_option("attn_ablate", ab)
    us = timeit(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, H, dq=dqkv[..., :D], dk=dqkv[..., D:2 * D], dv=dqkv[..., 2 * D:], parts=2))
    print(f"bwd dkv ablate={ab}: {us:7.1f} us")
ops.set_option("attn_ablate", 0)
for parts, name in ((1, "dq"), (2, "dkv"), (3, "both")):
    fn = lambda: ops.attn_bwd(q, k, v, o, d_o, lse, H, dq=dqkv[..., :D], dk=dqkv[..., D:2 * D], dv=dqkv[..., 2 * D:], parts=parts)
    us = timeit(fn)
    fl = {1: 6.0, 2: 8.0, 3: 14.0}[parts] * B * H * T * T * 64
    print(f"bwd {name:5s}: {us:7.1f} us  {fl / us / 1e6:6.0f} TFLOP/s")
