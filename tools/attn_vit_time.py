"""ViT-B/16 attention (B x 12 heads x 197 tokens x 64) forward / dQ / dK,dV kernels, event-timed; with
CFHIP_LIB=tools/libcfhip_ablate.so also the phase ablations (1: no K/V (Q/dO) DMA, 2: prologue only, 4: no stores)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cflearn_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T, H, D = 197, 12, 768
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, T, 3 * D, device=dev, generator=g).to(torch.bfloat16)
d_o = torch.randn(B, T, D, device=dev, generator=g).to(torch.bfloat16)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
o, lse = ops.attn_fwd(q, k, v, H)
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
delta = torch.empty(B, H, T, device=dev)


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


abl = [0] + ([1, 2, 4] if "ablate" in os.environ.get("CFHIP_LIB", "") else [])
if len(abl) == 1:
    abl = [("persistent", 1), ("short_max", 0), ("persistent", 1), ("short_max", 0)]
for a in abl:
    if isinstance(a, tuple):
        ops.set_option("attn_short_max", 256 if a[0] == "persistent" else a[1])  # short_max 0: the general two-tile kernels
        if a[0] == "persistent":
            ops.set_option("attn_persistent", a[1])
    elif a or len(abl) > 1:
        ops.set_option("attn_ablate", a)
    fw = t(lambda: ops.attn_fwd(q, k, v, H))
    b1 = t(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, H, dq=dq, dk=dk, dv=dv, parts=1, delta=delta))
    b2 = t(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, H, dq=dq, dk=dk, dv=dv, parts=2, delta=delta))
    b3 = t(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, H, dq=dq, dk=dk, dv=dv, parts=3, delta=delta))
    print(f"ablate {a}: fwd {fw:6.1f} us   dq {b1:6.1f}   dkv {b2:6.1f}   both (one call) {b3:6.1f}", flush=True)
