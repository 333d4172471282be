"""Tile walk order sweep (`gemm_group_n`: > 0 column groups, < 0 row groups, 0 = rows outer / all columns inner): isolated time
of every GEMM launch shape of the ViT-B/16 step per setting.   python tools/gemm_group_sweep.py [batch] [values]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cflearn_amd import ops
from tools.gemm_bench import make

dev = torch.device("cuda")
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 128
VALUES = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3,4,6,8,12,-1,-2,-4,-8,-16").split(",")]
shapes = [s for s in bench.gemm_shapes(BATCH) if s[2] > 1024 or s[1] == "tn"]
print("shape".ljust(34) + "".join(f"{v:>7d}" for v in VALUES))
for rnd in range(2):
    for count, layout, m, n, k, epi in shapes:
        if min(m, n) < 512:
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        a, b, bias, aux, out, kw = make(layout, m, n, k, epi, dev, g)
        if epi == "residual":
            out = torch.empty(m, n, dtype=torch.float32, device=dev)
            kw["aux_in"] = torch.randn(m, n, device=dev, generator=g)
        line = []
        for v in VALUES:
            ops.set_option("gemm_group_n", v)
            for _ in range(3):
                ops.gemm(a, b, bias=bias, out=out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(a, b, bias=bias, out=out, **kw)
            e1.record(); e1.synchronize()
            line.append(e0.elapsed_time(e1) * 50)
        ops.set_option("gemm_group_n", 0)
        print(f"{layout} {m:5d}x{n:4d}x{k:5d} {epi:8s} x{count:2d} " + "".join(f"{u:7.1f}" for u in line), flush=True)
