"""GEMM tile-configuration A/B on the ViT-B/16 shapes: correctness vs torch fp32 + event timing.

    python tools/gemm_bench.py [--batch 64] [--reps 20] [--configs 0,1,2,3]
Prints one line per (shape, config) and a per-config weighted total (ms of GEMM per training step).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from cflearn_amd import ops  # noqa: E402


def make(layout, m, n, k, epi, dev, g):
    bf = torch.bfloat16
    rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(bf)  # noqa: E731
    if layout == "nt":
        a, b, kw = rnd(m, k), rnd(n, k), {}
    elif layout == "nn":
        a, b, kw = rnd(m, k), rnd(k, n), dict(b_trans=True)
    else:
        a, b, kw = rnd(k, m), rnd(k, n), dict(a_trans=True, b_trans=True, split_k=ops.pick_split_k(m, n, k))
    out_dtype = torch.float32 if layout == "tn" else bf
    bias = torch.randn(n, device=dev, generator=g) if epi in ("bias", "residual", "gelu") else None
    aux = rnd(m, n) if epi in ("residual", "dgelu") else None
    if epi == "gelu":
        kw.update(epilogue=ops.EPI_GELU, aux_out=torch.empty(m, n, dtype=bf, device=dev))
    elif epi == "residual":
        kw.update(epilogue=ops.EPI_RESIDUAL, aux_in=aux)
    elif epi == "dgelu":
        kw.update(epilogue=ops.EPI_DGELU, aux_in=aux)
    out = torch.empty(m, n, dtype=out_dtype, device=dev)
    return a, b, bias, aux, out, kw


def reference(layout, a, b, bias, aux, epi, rows):
    A = a.float().t()[rows] if layout == "tn" else a.float()[rows]
    B = b.float() if layout in ("nn", "tn") else b.float().t()
    y = A @ B
    if bias is not None:
        y = y + bias
    if epi == "gelu":
        y = torch.nn.functional.gelu(y.to(torch.bfloat16).float())
    elif epi == "residual":
        y = y + aux.float()[rows]
    elif epi == "dgelu":
        x = aux.float()[rows].requires_grad_(True)
        torch.nn.functional.gelu(x).backward(torch.ones_like(x))
        y = y * x.grad
    return y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--configs", default="0,1,3,7,8")
    ap.add_argument("--lt", action="store_true", help="also time torch.matmul (hipBLASLt, no epilogue) for context")
    args = ap.parse_args()
    dev = torch.device("cuda")
    cfgs = [int(c) for c in args.configs.split(",")]
    shapes = [s for s in bench.gemm_shapes(args.batch, grouped=False) if s[2] > 64 and s[4] > 64]  # skip the tiny head GEMMs
    totals = {c: 0.0 for c in cfgs}
    best_total = 0.0
    flops_total = 0.0
    for count, layout, m, n, k, epi in shapes:
        g = torch.Generator(device=dev).manual_seed(m + n + k)
        a, b, bias, aux, out, kw = make(layout, m, n, k, epi, dev, g)
        rows = torch.randint(0, m, (64,), device=dev)
        want = reference(layout, a, b, bias, aux, epi, rows)
        line, best = [], None
        for c in cfgs:
            ops.set_option("gemm_config", c)
            out.zero_()
            ops.gemm(a, b, bias=bias, out=out, **kw)
            err = ((out.float()[rows] - want).norm() / want.norm()).item()
            ok = err < (2e-5 if out.dtype == torch.float32 else 6e-3)
            for _ in range(2):
                ops.gemm(a, b, bias=bias, out=out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                ops.gemm(a, b, bias=bias, out=out, **kw)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.reps
            tf = 2.0 * m * n * k / us / 1e6
            totals[c] += count * us
            best = us if best is None else min(best, us)
            line.append(f"c{c}: {us:7.1f}us {tf:6.0f}TF {'ok' if ok else f'BAD({err:.1e})'}")
        if args.lt:
            fn = {"nt": lambda: a @ b.t(), "nn": lambda: a @ b, "tn": lambda: a.t() @ b}[layout]
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.reps
            line.append(f"hipBLASLt(no epi): {us:7.1f}us {2.0 * m * n * k / us / 1e6:6.0f}TF")
        best_total += count * best
        flops_total += count * 2.0 * m * n * k
        print(f"{layout} {m:6d}x{n:5d}x{k:6d} {epi:9s} x{count:2d} | " + " | ".join(line), flush=True)
    for c in cfgs:
        print(f"config {c}: {totals[c] / 1e3:.3f} ms GEMM / step  ({flops_total / totals[c] / 1e6:.0f} TF avg)")
    print(f"best-of : {best_total / 1e3:.3f} ms GEMM / step  ({flops_total / best_total / 1e6:.0f} TF avg)")
    ops.set_option("gemm_config", -1)


if __name__ == "__main__":
    main()
