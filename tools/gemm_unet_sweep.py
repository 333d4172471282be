"""Tile configurations on the DDPM UNet's forward / dX GEMM shapes (64^2 x 8 step: profiles/r03/unet64_gemm_table.txt).
    python tools/gemm_unet_sweep.py [--reps 30]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402

SHAPES = [  # (layout, M, N, K, launches per step)
    ("nn", 2048, 1280, 1280, 50), ("nn", 32768, 320, 320, 50), ("nn", 8192, 640, 640, 50), ("nt", 2048, 1280, 1280, 50),
    ("nt", 32768, 320, 320, 50), ("nt", 8192, 640, 640, 50), ("nn", 2048, 1280, 10240, 5), ("nn", 32768, 320, 2560, 5),
    ("nn", 8192, 640, 5120, 5), ("nn", 512, 1280, 1280, 10), ("nt", 32768, 2560, 320, 5), ("nt", 2048, 1280, 5120, 5),
    ("nt", 8192, 5120, 640, 5), ("nt", 2048, 10240, 1280, 5), ("nt", 32768, 320, 1280, 5), ("nn", 32768, 640, 320, 2),
    ("nn", 32768, 960, 320, 1), ("nt", 8192, 640, 2560, 5), ("nn", 512, 1280, 10240, 1), ("nn", 32768, 1280, 320, 5),
    ("nn", 8192, 2560, 640, 5), ("nn", 2048, 5120, 1280, 5), ("nt", 512, 1280, 1280, 10), ("nt", 512, 1280, 5120, 1),
]
CFGS = (14, 15, 0, 1, 2, 3, 5, 6, 8)
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.randn(r, c, generator=g, device=dev) * 0.5).to(torch.bfloat16)  # noqa: E731
tot = {c: 0.0 for c in CFGS}
tot["heuristic"] = 0.0
tot["best"] = 0.0
print(f"{'shape':<26}" + "".join(f"{'c' + str(c):>8}" for c in CFGS) + "   heur   (us)")
for layout, m, n, k, cnt in SHAPES:
    a = rnd(m, k)
    b = rnd(n, k) if layout == "nt" else rnd(k, n)
    out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    res = {}
    for c in CFGS + (-1,):
        ops.set_option("gemm_config", c)
        try:
            for _ in range(3):
                ops.gemm(a, b, b_trans=(layout == "nn"), out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                ops.gemm(a, b, b_trans=(layout == "nn"), out=out)
            e1.record()
            e1.synchronize()
            res[c] = e0.elapsed_time(e1) * 1e3 / args.reps
        except RuntimeError:
            res[c] = float("nan")
    ops.set_option("gemm_config", -1)
    for c in CFGS:
        tot[c] += res[c] * cnt
    tot["heuristic"] += res[-1] * cnt
    best = min(v for v in res.values() if v == v)
    tot["best"] += best * cnt
    print(f"{layout} {m:>6}x{n:>5}x{k:>5} x{cnt:<3}" + "".join(f"{res[c]:8.1f}" for c in CFGS) + f"{res[-1]:8.1f}   best c{min((v, c) for c, v in res.items() if v == v and c >= 0)[1]}")
print("ms per step: " + "  ".join(f"{k}: {v / 1e3:.2f}" for k, v in tot.items()))
