"""L2-fill bytes per launch of the ViT-B/16 step's plain GEMM shapes under different tile walks (option gemm_group_n).

    tools/gemm_traffic.sh <tag> [--gn 8,0,12,...]        (on the GPU box; wraps this script in rocprofv3 --pmc FETCH_SIZE)
    python tools/gemm_traffic.py --labels-only            (prints the launch sequence the wrapper joins the counters to)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # (name, layout, M, N, K): one backward / forward slice of batch 128 (M = 64 x 197)
    ("qkv fwd", "nt", 12608, 2304, 768), ("proj fwd", "nt", 12608, 768, 768), ("fc1 fwd", "nt", 12608, 3072, 768),
    ("fc2 fwd", "nt", 12608, 768, 3072), ("qkv dX", "nn", 12608, 768, 2304), ("proj dX", "nn", 12608, 768, 768),
    ("fc2 dX", "nn", 12608, 3072, 768), ("fc1 dX", "nn", 12608, 768, 3072),
]
REPS = 3


def labels(gns):
    return [(name, gn, (m * k + n * k) * 2) for gn in gns for name, _, m, n, k in SHAPES for _ in range(REPS)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gn", default="8,0,4,12,16,24,-4,-8,-16")
    ap.add_argument("--labels-only", action="store_true")
    ap.add_argument("--join", default=None, help="csv of (kernel, value KiB) rows in dispatch order: print the table")
    args = ap.parse_args()
    gns = [int(v) for v in args.gn.split(",")]
    if args.join:
        import csv
        rows = [r for r in csv.reader(open(args.join)) if "gemm_bf16" in r[0]]
        lab = labels(gns)
        if len(rows) != len(lab):
            raise SystemExit(f"{len(rows)} GEMM dispatches in the trace, {len(lab)} expected")
        acc = {}
        for (name, gn, alg), r in zip(lab, rows):
            acc.setdefault((name, gn), []).append(float(r[1]) * 2048.0)
        names = [s[0] for s in SHAPES]
        print(f"{'MB of L2 fills per launch':<28}" + "".join(f"{'gn ' + str(g):>9}" for g in gns) + "   algorithmic")
        for name, _, m, n, k in SHAPES:
            print(f"{name + f' {m}x{n}x{k}':<28}" + "".join(f"{sum(acc[(name, g)]) / REPS / 1e6:9.1f}" for g in gns) + f"   {(m * k + n * k) * 2 / 1e6:9.1f}")
        return
    if args.labels_only:
        for row in labels(gns):
            print(*row)
        return
    import torch
    from cflearn_amd import ops

    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda r, c: (torch.randn(r, c, generator=g, device=dev) * 0.5).to(torch.bfloat16)  # noqa: E731
    ten = {}
    for name, layout, m, n, k in SHAPES:
        ten[name] = (rnd(m, k), rnd(n, k) if layout == "nt" else rnd(k, n), torch.empty(m, n, dtype=torch.bfloat16, device=dev))
    torch.cuda.synchronize()
    for gn in gns:
        ops.set_option("gemm_group_n", gn)
        for name, layout, m, n, k in SHAPES:
            a, b, out = ten[name]
            for _ in range(REPS):
                ops.gemm(a, b, b_trans=(layout == "nn"), out=out)
    torch.cuda.synchronize()


main()
