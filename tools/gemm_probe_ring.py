"""Probe (needs the -DCFHIP_ABLATE build: CFHIP_LIB=tools/libcfhip_ablate.so): can ONE workgroup per CU with a deep
LDS ring sustain the K loop that two co-resident workgroups with shallow rings reach?  Full launch | K loop only
(no epilogue) for the 256x128 two-group kernel with a 3-slot ring (2 WG / CU, cfg 8) and a 6-slot ring (1 WG / CU,
cfg 11), and the 256x256 kernel with 4 / 5 slots (cfg 7 / 12)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cflearn_amd import ops
from tools.gemm_bench import make

dev = torch.device("cuda")
M = 25216
shapes = [("nt", M, 3072, 768, "gelu"), ("nt", M, 768, 3072, "residual"), ("nt", M, 2304, 768, "bias"),
          ("nn", M, 3072, 768, "dgelu"), ("nn", M, 768, 3072, "none"), ("nt", M, 768, 768, "residual")]
for layout, m, n, k, epi in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    a, b, bias, aux, out, kw = make(layout, m, n, k, epi, dev, g)
    line = []
    for c in (8, 11, 7, 12):
        ops.set_option("gemm_config", c)
        res = []
        for ab in (0, 4):
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
        line.append(f"cfg{c}: {res[0]:6.1f} / {res[1]:6.1f}")
    print(f"{layout} {m}x{n}x{k} {epi:8s} full / K-loop only (us) | " + " | ".join(line), flush=True)
