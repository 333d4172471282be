"""Tile configurations on the CLIP text tower's forward / dX GEMM shapes (M = 256 x 77 = 19 712 rows, 512 wide), with the
epilogues of the model (bias, bias + f32 residual, quick-GELU, quick-GELU'), hot (0.3 s per row):
    python tools/gemm_clip_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402

M = 256 * 77
SHAPES = [  # (layout, N, K, epilogue, launches per step: 12 layers, one pipeline per tower)
    ("nt", 1536, 512, "bias", 12), ("nt", 512, 512, "residual", 12), ("nt", 2048, 512, "qgelu", 12), ("nt", 512, 2048, "residual", 12),
    ("nn", 2048, 512, "dqgelu", 12), ("nn", 512, 2048, "none", 12), ("nn", 512, 512, "none", 12), ("nn", 512, 1536, "none", 12),
]
CFGS = (15, 14, 0, 3, 13, 16)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16
rnd = lambda r, c: (torch.randn(r, c, generator=g, device=dev) * 0.5).to(bf)  # noqa: E731
tot = {c: 0.0 for c in CFGS}
tot["heuristic"] = tot["best"] = 0.0
print(f"{'shape':<34}" + "".join(f"{'c' + str(c):>8}" for c in CFGS) + "   heur   (us)")
for layout, n, k, epi, cnt in SHAPES:
    a = rnd(M, k)
    b = rnd(n, k) if layout == "nt" else rnd(k, n)
    kw = dict(b_trans=(layout == "nn"))
    bias = torch.zeros(n, device=dev) if epi in ("bias", "residual", "qgelu") else None
    odt = bf
    if epi == "residual":
        kw.update(epilogue=ops.EPI_RESIDUAL, aux_in=torch.randn(M, n, generator=g, device=dev))
        odt = torch.float32
    elif epi == "qgelu":
        kw.update(epilogue=ops.EPI_QGELU, aux_out=torch.empty(M, n, dtype=bf, device=dev))
    elif epi == "dqgelu":
        kw.update(epilogue=ops.EPI_DQGELU, aux_in=rnd(M, n))
    out = torch.empty(M, n, dtype=odt, device=dev)
    res = {}
    for c in CFGS + (-1,):
        ops.set_option("gemm_config", c)
        try:
            fn = lambda: ops.gemm(a, b, bias=bias, out=out, **kw)  # noqa: E731
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps, t0 = 0, time.perf_counter()
            e0.record()
            while time.perf_counter() - t0 < 0.3:
                for _ in range(20):
                    fn()
                reps += 20
                torch.cuda.synchronize()
            e1.record()
            e1.synchronize()
            res[c] = e0.elapsed_time(e1) * 1e3 / reps
        except RuntimeError:
            res[c] = float("nan")
    ops.set_option("gemm_config", -1)
    for c in CFGS:
        tot[c] += res[c] * cnt
    tot["heuristic"] += res[-1] * cnt
    best = min(v for v in res.values() if v == v)
    tot["best"] += best * cnt
    print(f"{layout} {M}x{n:>5}x{k:>5} {epi:<8} x{cnt:<3}" + "".join(f"{res[c]:8.1f}" for c in CFGS) + f"{res[-1]:8.1f}   best c{min((v, c) for c, v in res.items() if v == v and c >= 0)[1]}"
          f"  {2.0 * M * n * k / best / 1e6:6.0f} TF")
print("ms per step: " + "  ".join(f"{k}: {v / 1e3:.2f}" for k, v in tot.items()))
