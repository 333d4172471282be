"""One GEMM shape launched repeatedly (for rocprofv3 --pmc passes): python tools/gemm_one.py nt 12608 768 3072 residual [config]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cflearn_amd import ops
from tools.gemm_bench import make
layout, m, n, k, epi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
if len(sys.argv) > 6:  # tile configuration (cfhip_set_option "gemm_config")
    ops.set_option("gemm_config", int(sys.argv[6]))
dev = torch.device("cuda")
a, b, bias, aux, out, kw = make(layout, m, n, k, epi, dev, torch.Generator(device=dev).manual_seed(1))
for _ in range(10):
    ops.gemm(a, b, bias=bias, out=out, **kw)
torch.cuda.synchronize()
