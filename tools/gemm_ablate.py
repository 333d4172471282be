"""Ablation of the GEMM kernel on a few ViT shapes: which resource bounds it?
ablate bits: 1 = no in-loop DMA, 2 = no MFMA / LDS reads, 4 = no epilogue stores."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cflearn_amd import ops
from tools.gemm_bench import make

dev = torch.device("cuda")
shapes = [("nt", 12608, 2304, 768, "bias"), ("nt", 12608, 768, 3072, "residual"), ("nt", 12608, 3072, 768, "gelu"),
          ("nn", 12608, 768, 2304, "none"), ("tn", 3072, 768, 12608, "none")]
for layout, m, n, k, epi in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    a, b, bias, aux, out, kw = make(layout, m, n, k, epi, dev, g)
    for c in (0, 2, 3):
        ops.set_option("gemm_config", c)
        res = []
        for ab in (0, 1, 2, 3, 4, 5, 6):
            ops.set_option("gemm_ablate", ab)
            for _ in range(2):
                ops.gemm(a, b, bias=bias, out=out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(a, b, bias=bias, out=out, **kw)
            e1.record(); e1.synchronize()
            res.append(e0.elapsed_time(e1) * 100)
        ops.set_option("gemm_ablate", 0)
        print(f"{layout} {m}x{n}x{k} {epi:8s} cfg{c}: full {res[0]:6.1f} | noDMA {res[1]:6.1f} | noMFMA {res[2]:6.1f} | "
              f"noDMA+noMFMA {res[3]:6.1f} | noStore {res[4]:6.1f} | noDMA+noStore {res[5]:6.1f} | noMFMA+noStore {res[6]:6.1f}  (us)", flush=True)
