"""Where the one-pass attention backward (8 waves, two key tiles per wave) spends a head: the launch with parts of it switched off.
    tools/build_variant.sh ablate -DCFHIP_ABLATE; CFHIP_LIB=tools/libcfhip_ablate.so python tools/one_pass_ablate.py [batch=128]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402

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
w = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
for _ in range(100):
    w @ w


def t(n=300):
    fn = lambda: ops.attn_bwd(q, k, v, o, d_o, lse, H, dq=dq, dk=dk, dv=dv, parts=3, delta=delta)  # noqa: E731
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


names = {0: "everything", 8: "no phase 1", 16: "no phase 2", 24: "no phase 1, 2", 32: "no statistics", 64: "no dK / dV stores", 128: "no K / V fragments",
         256: "no staging of the next head", 8 + 16 + 32 + 64 + 128 + 256: "nothing but barriers", 32 + 64 + 128 + 256: "phases only", 8 + 16 + 32: "staging + fragments + stores"}
for one in (2, 1):
    ops.set_option("attn_one_pass", one)
    print("8 waves, two key tiles" if one == 2 else "16 waves (no ablation hooks: reference)")
    for a in (names if one == 2 else [0]):
        ops.set_option("attn_ablate", a)
        print(f"  {names[a]:32s} {t():7.1f} us", flush=True)
ops.set_option("attn_ablate", 0)
