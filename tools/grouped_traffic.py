"""The grouped weight-gradient launch of two ViT-B/16 blocks (batch 128), alone: for tools/pmc_fetch.sh.
    python tools/grouped_traffic.py [--bias 0|1] [--reps 4] [--variant V]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bias", type=int, default=1)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--blocks", type=int, default=2)
ap.add_argument("--variant", type=int, default=0)
args = ap.parse_args()
ops.set_option("grouped_variant", args.variant)
dev = torch.device("cuda")
k = 128 * 197
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda r, c: (torch.randn(r, c, generator=g, device=dev) * 0.5).to(torch.bfloat16)  # noqa: E731
shapes = ((2304, 768), (768, 768), (3072, 768), (768, 3072))  # backward order does not matter here: qkv, proj, fc1, fc2
probs = []
for _ in range(args.blocks):
    for m, n in shapes:
        probs.append((rnd(k, m), rnd(k, n), torch.empty(m, n, dtype=torch.float32, device=dev), False,
                      torch.empty(m, dtype=torch.float32, device=dev) if args.bias else None, False))
alg = sum((dy.numel() + x.numel()) * 2 for dy, x, *_ in probs)
for _ in range(2):
    ops.gemm_grouped_tn(probs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps):
    ops.gemm_grouped_tn(probs)
e1.record()
e1.synchronize()
us = e0.elapsed_time(e1) * 1e3 / args.reps
fl = sum(2.0 * dy.shape[1] * x.shape[1] * k for dy, x, *_ in probs)
print(f"bias {args.bias} variant {args.variant}: {us:.1f} us / launch, {fl / us / 1e6:.0f} TFLOP/s; algorithmic operand bytes {alg / 1e6:.1f} MB")
