"""Weight-gradient GEMMs of ViT-B/16 blocks: one split-K GEMM + reduce per gradient (round 1/2) vs the grouped launch.

    python tools/gemm_grouped_bench.py [--batch 128] [--reps 10]
Times, per transformer block: (a) the four ops.gemm(a_trans, b_trans, split_k, bias_grad) launches; (b) ops.gemm_grouped_tn
over 1 / 2 / 3 blocks per launch, every ring variant."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--configs", default="1,7,14")
    ap.add_argument("--variants", default="0,1,2,3,4")
    args = ap.parse_args()
    dev = torch.device("cuda")
    k = args.batch * 197
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda r, c: (torch.randn(r, c, generator=g, device=dev) * 0.5).to(torch.bfloat16)  # noqa: E731
    shapes = ((768, 3072), (3072, 768), (768, 768), (2304, 768))
    blocks = []
    for _ in range(3):
        blocks.append([(rnd(k, m), rnd(k, n), torch.empty(m, n, dtype=torch.float32, device=dev), False,
                        torch.empty(m, dtype=torch.float32, device=dev), False) for m, n in shapes])
    flops_block = sum(2.0 * m * n * k for m, n in shapes)

    state = dict(bias=True, split=None)

    def old_path():
        for dy, x, out, _, bg, _ in blocks[0]:
            split = state["split"] or ops.pick_split_k(out.shape[0], out.shape[1], k)
            ops.gemm(dy, x, a_trans=True, b_trans=True, out=out, split_k=split, bias_grad=bg if state["bias"] else None)

    for cfg in [int(c) for c in args.configs.split(",")]:
        ops.set_option("gemm_config", cfg)
        state["bias"] = cfg not in (7, 8, 9, 10, 11, 12)  # the phase kernel has no fused bias gradient
        us = timed(old_path, args.reps)
        print(f"per-GEMM split-K + reduce, gemm_config {cfg:2d}: {us:8.1f} us / block  {flops_block / us / 1e6:6.0f} TF", flush=True)
    ops.set_option("gemm_config", 7)  # 256 x 256 x 32 phase kernel at the split that fills 256 CUs with 256-wide tiles
    state["bias"] = False
    for split in (4, 7):
        state["split"] = split
        us = timed(old_path, args.reps)
        print(f"per-GEMM split-K + reduce, gemm_config  7, split {split}: {us:8.1f} us / block  {flops_block / us / 1e6:6.0f} TF", flush=True)
    state.update(bias=True, split=None)
    ops.set_option("gemm_config", -1)
    us = timed(old_path, args.reps)
    print(f"per-GEMM split-K + reduce, heuristic     : {us:8.1f} us / block  {flops_block / us / 1e6:6.0f} TF", flush=True)
    for variant in [int(v) for v in args.variants.split(",")]:
        ops.set_option("grouped_variant", variant)
        for nb in (1, 2, 3):
            probs = [p for b in blocks[:nb] for p in b]
            us = timed(lambda: ops.gemm_grouped_tn(probs), args.reps)
            tiles = nb * sum((m // 256) * (n // 256) for m, n in shapes)
            print(f"grouped variant {variant}, {nb} block(s) / launch ({tiles:3d} tiles): {us:8.1f} us = {us / nb:8.1f} us / block  "
                  f"{nb * flops_block / us / 1e6:6.0f} TF", flush=True)
        nobias = [(dy, x, out, acc, None, False) for b in blocks[:2] for (dy, x, out, acc, _, _) in b]
        us = timed(lambda: ops.gemm_grouped_tn(nobias), args.reps)
        print(f"grouped variant {variant}, 2 blocks, no bias gradients    : {us:8.1f} us = {us / 2:8.1f} us / block  "
              f"{2 * flops_block / us / 1e6:6.0f} TF", flush=True)
    ops.set_option("grouped_variant", 0)


if __name__ == "__main__":
    main()
