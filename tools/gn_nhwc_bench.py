"""GroupNorm on NHWC rows: the group form (one workgroup per (sample, group), one launch) against the slice form (row slices, three
launches) on the zoo UNet's shapes at batch 8.      python tools/gn_nhwc_bench.py [batch]
Each case rotates over 4 operand sets (the tensors of a step come from the previous kernel, not from a warm L2)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
SHAPES = [(4096, 320), (4096, 640), (4096, 960), (1024, 320), (1024, 640), (1024, 960), (1024, 1280), (1024, 1920),
          (256, 640), (256, 1280), (256, 1920), (256, 2560), (64, 1280), (64, 2560)]
FORMS = [("group", None), ("slices/target", "default"), ("slices 8", 8), ("slices 16", 16), ("slices 32", 32), ("slices 64", 64)]


def timed(fn, reps=20):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


orig = ops.gn_nhwc_splits
print(f"batch {B}; us per call: forward (SiLU, time-embedding add) / backward")
for inner, c in SHAPES:
    sets = []
    for k in range(4):
        x = torch.randn(B * inner, c, device=dev).bfloat16()
        dy = torch.randn(B * inner, c, device=dev).bfloat16()
        sets.append((x, dy))
    gamma = torch.randn(c, device=dev) * 0.2 + 1.0
    beta = torch.randn(c, device=dev) * 0.2
    add = torch.randn(B, c, device=dev) * 0.3
    cells = []
    for name, sp in FORMS:
        if sp is None:
            ops.gn_nhwc_splits = lambda b, i, cc=0, g=0, bw=False: 0
        elif sp == "default":
            ops.gn_nhwc_splits = lambda b, i, cc=0, g=0, bw=False: max(1, min(-(-ops.GN_NHWC_TARGET_WORKGROUPS // b), i // ops.GN_NHWC_MIN_ROWS, 4096))
        else:
            if inner // sp < 4:
                cells.append(f"{name}: -")
                continue
            ops.gn_nhwc_splits = lambda b, i, cc=0, g=0, bw=False, sp=sp: sp
        try:
            stats = ops.groupnorm_nhwc_fwd(sets[0][0], B, gamma, beta, 32, 1e-6, add=add, silu=True)
            tf = timed(lambda i: ops.groupnorm_nhwc_fwd(sets[i % 4][0], B, gamma, beta, 32, 1e-6, add=add, silu=True))
            tb = timed(lambda i: ops.groupnorm_nhwc_bwd(sets[i % 4][1], sets[i % 4][0], B, gamma, beta, stats[1], stats[2], 32, add=add, silu=True))
            cells.append(f"{name}: {tf:6.1f} / {tb:6.1f}")
        finally:
            ops.gn_nhwc_splits = orig
    mb = B * inner * c * 2 / 1e6
    print(f"rows {inner:5d} x C {c:5d} ({mb:5.1f} MB): " + " | ".join(cells), flush=True)
