"""Weight-gradient (tn) GEMMs with the WHOLE K = batch x tokens reduction per tile (no split-K), one full round of tiles per
configuration: the K-loop rate of each tile configuration on the dW layout (both operands m-major, `ds_read_b64_tr_b16`),
next to the 256x256x32 grouped kernel (160 KB of LDS, one workgroup per CU).

    python tools/tn_whole_k_probe.py [K]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402

dev = torch.device("cuda")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 25216
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(torch.bfloat16)  # noqa: E731


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ref = None
for cfg, (m, n), what in ((15, (3072, 4096), "192x128x64, 4 waves, 2 WG/CU: 512 tiles"), (14, (3072, 4096), "192x128x64, 8 waves, 2 WG/CU: 512 tiles"),
                          (0, (4096, 2048), "128x128x64, 2 WG/CU: 512 tiles"), (1, (4096, 4096), "128x128x32, 4 WG/CU: 1024 tiles"),
                          (13, (4096, 4096), "256x256x64, 1 WG/CU: 256 tiles"), (15, (3072, 2304), "192x128x64, 4 waves: 288 tiles (one ViT block)"),
                          (15, (4608, 3072), "192x128x64, 4 waves: 576 tiles (two ViT blocks)")):
    a, b = rnd(K, m), rnd(K, n)
    out = torch.empty(m, n, dtype=torch.float32, device=dev)
    ops.set_option("gemm_config", cfg)
    us = timed(lambda: ops.gemm(a, b, a_trans=True, b_trans=True, out=out, split_k=1))
    ops.set_option("gemm_config", -1)
    rows = torch.randint(0, m, (16,), device=dev)
    want = a.float().t()[rows] @ b.float()
    err = ((out[rows] - want).norm() / want.norm()).item()
    print(f"c{cfg:<2d} {m}x{n}x{K}  {what:58s} {us:8.1f} us  {2.0 * m * n * K / us / 1e6:7.1f} TFLOP/s   rel {err:.1e}")
    del a, b, out
m = n = 4096
a, b = rnd(K, m), rnd(K, n)
out = torch.empty(m, n, dtype=torch.float32, device=dev)
us = timed(lambda: ops.gemm_grouped_tn([(a, b, out, False, None, False)]))
print(f"grouped 256x256x32, 160 KB, 1 WG/CU: 256 tiles  {m}x{n}x{K}  {us:8.1f} us  {2.0 * m * n * K / us / 1e6:7.1f} TFLOP/s")
