"""Epilogue cost split at M = 25216: full | no epilogue at all (4) | epilogue math without the stores (8)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cflearn_amd import ops
from tools.gemm_bench import make

dev = torch.device("cuda")
M = 25216
shapes = [("nt", M, 2304, 768, "bias"), ("nt", M, 3072, 768, "gelu"), ("nt", M, 768, 3072, "residual"),
          ("nn", M, 3072, 768, "dgelu"), ("nn", M, 768, 3072, "none"), ("tn", 768, 3072, M, "none")]
for layout, m, n, k, epi in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    a, b, bias, aux, out, kw = make(layout, m, n, k, epi, dev, g)
    for c in (0, 1, 8):
        ops.set_option("gemm_config", c)
        res = []
        for ab in (0, 4, 8, 1, 2):
            if c == 8 and ab in (1, 2):
                res.append(float("nan")); continue
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
        print(f"{layout} {m}x{n}x{k} {epi:8s} cfg{c}: full {res[0]:6.1f} | no epilogue {res[1]:6.1f} | epilogue math, no stores {res[2]:6.1f} | noDMA {res[3]:6.1f} | noMFMA {res[4]:6.1f} (us)", flush=True)
