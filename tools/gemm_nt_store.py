import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cflearn_amd import ops
from tools.gemm_bench import make
dev = torch.device("cuda")
for layout, m, n, k, epi in [("nt", 12608, 2304, 768, "bias"), ("nt", 12608, 3072, 768, "gelu"), ("nt", 12608, 768, 3072, "residual"), ("nn", 12608, 3072, 768, "dgelu")]:
    g = torch.Generator(device=dev).manual_seed(1)
    a, b, bias, aux, out, kw = make(layout, m, n, k, epi, dev, g)
    res = []
    for ab in (0, 8):
        ops.set_option("gemm_ablate", ab)
        for _ in range(3): ops.gemm(a, b, bias=bias, out=out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(a, b, bias=bias, out=out, **kw)
        e1.record(); e1.synchronize()
        res.append(e0.elapsed_time(e1) * 50)
    ops.set_option("gemm_ablate", 0)
    print(f"{layout} {m}x{n}x{k} {epi}: normal {res[0]:.1f} us | nontemporal stores {res[1]:.1f} us")
