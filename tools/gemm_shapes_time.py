"""Event-timed duration of every GEMM launch shape of the ViT-B/16 step in the MODEL's forms (f32 residual
stream, fused GELU / GELU' epilogues, split-K dW): `bench.time_gemms`, one line per shape.

    [CFHIP_LIB=tools/libcfhip_<variant>.so] python tools/gemm_shapes_time.py [--batch 128] [--reps 20]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
flops, tsec, rows, nbytes = bench.time_gemms(a.batch, a.reps)
for r in rows:
    print(f"{r['layout']} {r['M']:6d}x{r['N']:5d}x{r['K']:6d} {r['epilogue']:9s} x{r['count']:2d}  {r['us']:8.1f} us  {r['tflops']:7.1f} TF")
print(f"lib={os.environ.get('CFHIP_LIB', 'default')}: {tsec * 1e3:.3f} ms GEMM / step, {flops / tsec / 1e12:.1f} TFLOP/s")
