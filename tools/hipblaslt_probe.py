import torch, time
dev="cuda"
def timed(fn,reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
g=torch.Generator(device=dev).manual_seed(0)
rnd=lambda *s:(torch.randn(*s,generator=g,device=dev)*0.5).to(torch.bfloat16)
for m,n,k in ((4096,4096,4096),(8192,8192,8192),(12608,768,3072),(12608,3072,768),(12608,2304,768),(25216,768,3072)):
    a,b=rnd(m,k),rnd(n,k)
    us=timed(lambda: torch.matmul(a,b.t()))
    bt=rnd(k,n)
    us2=timed(lambda: torch.matmul(a,bt))
    print(f"hipBLASLt nt {m}x{n}x{k}: {us:8.1f} us {2*m*n*k/us/1e6:7.1f} TF | nn {us2:8.1f} us {2*m*n*k/us2/1e6:7.1f} TF")
    z=torch.zeros_like(a); zb=torch.zeros_like(b)
    us3=timed(lambda: torch.matmul(z,zb.t()))
    print(f"      zero operands nt: {us3:8.1f} us {2*m*n*k/us3/1e6:7.1f} TF")
k=25216
for m,n in ((4096,4096),(3072,768),(768,3072),(2304,768)):
    a,b=rnd(k,m),rnd(k,n)
    us=timed(lambda: torch.matmul(a.t(),b))
    print(f"hipBLASLt tn {m}x{n}x{k}: {us:8.1f} us {2*m*n*k/us/1e6:7.1f} TF")

# the same shapes on this package's kernels, random and ZERO operands: separates the K loop's structure from what the chip's power
# management does to a kernel that toggles every operand bit
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cflearn_amd import ops  # noqa: E402
for m, n, k in ((8192, 8192, 8192), (4096, 4096, 4096), (12608, 768, 3072), (12608, 3072, 768)):
    a, b = rnd(m, k), rnd(n, k)
    z, zb = torch.zeros_like(a), torch.zeros_like(b)
    out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    line = [f"cfhip nt {m}x{n}x{k}:"]
    for cfg in (13, 15, 14, 18):
        ops.set_option("gemm_config", cfg)
        us = timed(lambda: ops.gemm(a, b, out=out))
        usz = timed(lambda: ops.gemm(z, zb, out=out))
        line.append(f"c{cfg} {2*m*n*k/us/1e6:6.0f} / zeros {2*m*n*k/usz/1e6:6.0f} TF")
    ops.set_option("gemm_config", -1)
    print("  |  ".join(line))
