"""Where an epilogue's time goes (needs CFHIP_LIB=tools/libcfhip_ablate.so): full launch | K loop only (ablate 4) |
K loop + epilogue math without its stores (ablate 8), per shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cflearn_amd import ops
from tools.gemm_bench import make

dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25216
shapes = [("nt", M, 3072, 768, "gelu"), ("nn", M, 3072, 768, "dgelu"), ("nt", M, 768, 3072, "residual"),
          ("nt", M, 2304, 768, "bias"), ("nn", M, 768, 3072, "none")]
for rnd in range(2):
    for layout, m, n, k, epi in shapes:
        g = torch.Generator(device=dev).manual_seed(1)
        a, b, bias, aux, out, kw = make(layout, m, n, k, epi, dev, g)
        if epi == "residual":
            out = torch.empty(m, n, dtype=torch.float32, device=dev)
            kw["aux_in"] = torch.randn(m, n, device=dev, generator=g)
        res = []
        for ab in (0, 4, 8):
            ops.set_option("gemm_ablate", ab)
            for _ in range(3):
                ops.gemm(a, b, bias=bias, out=out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(a, b, bias=bias, out=out, **kw)
            e1.record(); e1.synchronize()
            res.append(e0.elapsed_time(e1) * 50)
        ops.set_option("gemm_ablate", 0)
        print(f"{layout} {m}x{n}x{k} {epi:8s} full {res[0]:6.1f} | K loop only {res[1]:6.1f} | + math, no stores {res[2]:6.1f}  (us)", flush=True)
