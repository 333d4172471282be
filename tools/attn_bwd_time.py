"""Event-timed attention forward / backward of the general kernels: python tools/attn_bwd_time.py [dh] [T] [B*H]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cflearn_amd import ops

dh = int(sys.argv[1]) if len(sys.argv) > 1 else 40
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
h = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
q, k, v, d_o = (torch.randn(1, T, h * dh, device=dev, generator=g).to(torch.bfloat16) for _ in range(4))
o, lse = ops.attn_fwd(q, k, v, h, head_dim=dh)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)


def t(fn, n=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


fw = t(lambda: ops.attn_fwd(q, k, v, h, head_dim=dh))
b1 = t(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, h, dq=dq, dk=dk, dv=dv, head_dim=dh, parts=1))
b2 = t(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, h, dq=dq, dk=dk, dv=dv, head_dim=dh, parts=2))
fl = 4.0 * h * T * T * dh
print(f"lib={os.environ.get('CFHIP_LIB', 'default')} dh={dh} T={T} H={h}: fwd {fw:.2f} ms ({fl / fw / 1e9:.0f} TF)  dq {b1:.2f} ms  dkv {b2:.2f} ms")
