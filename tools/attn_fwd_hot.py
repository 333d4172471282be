"""ViT-B/16 attention forward (B x 12 heads x 197 x 64, persistent form) hot: >= 1 s of back-to-back launches, us per launch.
    [CFHIP_LIB=tools/libcfhip_<variant>.so] python tools/attn_fwd_hot.py [batch=64]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, H, D = 197, 12, 768
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, T, 3 * D, device=dev, generator=g).to(torch.bfloat16)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
o, lse = ops.attn_fwd(q, k, v, H)
ref = o.clone()
for rnd in range(3):
    n = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        for _ in range(200):
            ops.attn_fwd(q, k, v, H)
        n += 200
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"batch {B}: {dt / n * 1e6:7.2f} us per forward", flush=True)
o2, lse2 = ops.attn_fwd(q, k, v, H)
print("checksum", o2.float().sum().item(), lse2.sum().item())
