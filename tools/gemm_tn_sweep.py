"""Weight-gradient (tn) GEMMs of the DDPM UNet: split-K depth x tile configuration, alone.   python tools/gemm_tn_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402

SHAPES = [(320, 320, 32768, 50), (640, 640, 8192, 50), (1280, 1280, 2048, 50), (5120, 640, 8192, 5), (2560, 320, 32768, 5),
          (320, 1280, 32768, 5), (10240, 1280, 2048, 5), (1280, 5120, 2048, 5), (640, 2560, 8192, 5), (1280, 1280, 512, 10)]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.randn(r, c, generator=g, device=dev) * 0.5).to(torch.bfloat16)  # noqa: E731


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for m, n, k, cnt in SHAPES:
    dy, x = rnd(k, m), rnd(k, n)
    out = torch.empty(m, n, dtype=torch.float32, device=dev)
    bg = torch.empty(m, dtype=torch.float32, device=dev)
    auto = ops.pick_split_k(m, n, k)
    row = []
    for cfg in (-1, 1, 0, 3, 14, 15):
        ops.set_option("gemm_config", cfg)
        for sk in sorted({auto, 1, 4, 8, 16, 32, 64}):
            if sk > max(1, k // 256):
                continue
            try:
                t = timed(lambda: ops.gemm(dy, x, a_trans=True, b_trans=True, out=out, split_k=sk, bias_grad=bg))
            except RuntimeError:
                continue
            row.append((t, cfg, sk))
    ops.set_option("gemm_config", -1)
    base = [t for t, c, s in row if c == -1 and s == auto][0]
    row.sort()
    print(f"tn {m:>5}x{n:>5}x{k:>6} x{cnt:<3} heuristic (split {auto:>3}) {base:7.1f} us | best: " +
          "  ".join(f"c{c} s{s} {t:.1f}" for t, c, s in row[:5]))
