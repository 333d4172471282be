"""One attention shape, a few launches of one pass (for rocprofv3 --pmc runs): python tools/attn_one.py B H T dh fwd|dq|dkv [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402

b, h, t, dh = (int(x) for x in sys.argv[1:5])
what = sys.argv[5]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
d = h * dh
rnd = lambda: (torch.randn(b, t, d, device=dev, generator=g) * 0.5).to(torch.bfloat16)  # noqa: E731
q, k, v, d_o = rnd(), rnd(), rnd(), rnd()
o, lse = ops.attn_fwd(q, k, v, h, head_dim=dh)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
delta = torch.empty(b, h, t, device=dev)
ops.attn_bwd(q, k, v, o, d_o, lse, h, dq=dq, dk=dk, dv=dv, parts=1, delta=delta, head_dim=dh)
for _ in range(reps):
    if what == "fwd":
        ops.attn_fwd(q, k, v, h, head_dim=dh)
    else:
        ops.attn_bwd(q, k, v, o, d_o, lse, h, dq=dq, dk=dk, dv=dv, parts=1 if what == "dq" else 2, delta=delta, head_dim=dh)
torch.cuda.synchronize()
