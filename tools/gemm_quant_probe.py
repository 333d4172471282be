"""Tile-quantisation probe: the N = 768 GEMMs of the step at row counts that give 510 / 516 / 594 / 1020 / 1182 tiles of
256 x 128 on the 512 resident workgroup slots (2 per CU).  If a launch is bound by whole 'rounds' of tiles, time does not
follow the row count.   python tools/gemm_quant_probe.py [config]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cflearn_amd import ops
from tools.gemm_bench import make

dev = torch.device("cuda")
cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8"])]
MS = [256 * 64, 256 * 85, 256 * 86, 256 * 92, 25216, 256 * 128, 256 * 170, 2 * 25216]
shapes = [("nt", 768, 3072, "residual"), ("nn", 768, 3072, "none"), ("nt", 768, 768, "residual"), ("nn", 768, 2304, "none"),
          ("nt", 3072, 768, "gelu")]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for c in cfgs:
    ops.set_option("gemm_config", c)
    for layout, n, k, epi in shapes:
        line = []
        for m in MS:
            g = torch.Generator(device=dev).manual_seed(1)
            a, b, bias, aux, out, kw = make(layout, m, n, k, epi, dev, g)
            if epi == "residual":  # the model's form: f32 residual stream
                out = torch.empty(m, n, dtype=torch.float32, device=dev)
                kw["aux_in"] = torch.randn(m, n, device=dev, generator=g)
            us = timeit(lambda: ops.gemm(a, b, bias=bias, out=out, **kw))
            line.append(f"M={m}: {us:6.1f}us {2.0 * m * n * k / us / 1e6:5.0f}TF")
        print(f"cfg{c} {layout} N={n} K={k} {epi:8s} | " + " | ".join(line), flush=True)
