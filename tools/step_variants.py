"""Within-process A/B of whole-step variants of the ViT-B/16 training step (interleaved rounds, medians).

    python tools/step_variants.py [batch] [name=python-expression ...]
Built-in variants exercise the round-3 switches: fused.DW_GROUP_BLOCKS / DW_GROUP_ON_MAIN, ring variants."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cflearn_amd as C  # noqa: E402
from cflearn_amd import fused, ops  # noqa: E402
from cflearn_amd.engine import TrainStep  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model = C.vit_b16_classifier(1000).to(dev)
ts = TrainStep(model, lr=1e-4, use_graph=False)
g = torch.Generator().manual_seed(1234)
img = torch.randn(BATCH, 3, 224, 224, generator=g).to(dev)
labels = torch.randint(0, 1000, (BATCH,), generator=g).to(dev)


def setv(blocks=2, on_main=False, variant=0, halves=2, bhalves=2, nt_wide=-1, nt=-1, nn=-1, share=0.5, heuristic=9, tn=-1, split=0):
    def f():
        ops.FORCE_SPLIT_K = split
        ops.set_option("attn_persistent", 7)
        ops.set_option("attn_pers_ctas", 256)
        ops.set_option("gemm_cfg_tn", tn)
        fused.FIRST_SLICE_SHARE = share
        ops.set_option("gemm_heuristic", heuristic)
        fused.DW_GROUP_BLOCKS = blocks
        fused.DW_GROUP_TILES = 0  # the variants of this tool group by block count
        fused.DW_PEROP_TAIL = 0
        fused.DW_GROUP_ON_MAIN = on_main
        fused.FWD_HALVES = halves
        fused.BWD_HALVES = bhalves
        ops.set_option("grouped_variant", variant)
        ops.set_option("gemm_group_n", 8)
        ops.set_option("gemm_cfg_nt_wide", nt_wide)
        ops.set_option("gemm_cfg_nt", nt)
        ops.set_option("gemm_cfg_nn", nn)
    return f


VARIANTS = {
    "default (g2, s2, table 9)": setv(2),
    "r4 table 8 (eight waves above N = 1024)": setv(2, heuristic=8),
    "heuristic 7 (192x128 on eight waves everywhere)": setv(2, heuristic=7),
    "nn=c15 (all dX on four waves)": setv(2, nn=15),
    "nn=c15 nt=c15": setv(2, nn=15, nt=15),
    # round 5: isolated sweep of the final r4 kernels (profiles/r05/gemm_bench_wide_tiles.log) has the four-wave 192x128x64 form ahead on
    # EVERY shape of the step, the wide GELU / GELU' outputs included (80.7 / 81.8 us against 99.4 / 105.4 on eight waves)
    "r5 all four-wave: ntw=c15 nt=c15 nn=c15": setv(2, nt_wide=15, nt=15, nn=15),
    "r5 ntw=c15 (FF1 + GELU on four waves)": setv(2, nt_wide=15),
    "r5 ntw=c15 nn=c15 (FF1 + GELU and GELU' on four waves)": setv(2, nt_wide=15, nn=15),
    "first slice 60 of 128": setv(2, share=60 / 128),
    "first slice 56 of 128": setv(2, share=56 / 128),
    "first slice 52 of 128": setv(2, share=52 / 128),
    "first slice 68 of 128": setv(2, share=68 / 128),
    "bias gradients by the first tile column (r3a)": setv(2, variant=16),
    "row-major tiles (r3a)": setv(2, variant=32),
    "r3a grouped kernel behaviour": setv(2, variant=48),
    "nn=c13": setv(2, nn=13),
    "nn=c13 ntw=c13": setv(2, nn=13, nt_wide=13),
    "nn=c13 ntw=c13 nt=c13": setv(2, nn=13, nt_wide=13, nt=13),
    "nn=c13 nt=c13": setv(2, nn=13, nt=13),
    "nn=c0 ntw=c0 nt=c0": setv(2, nn=0, nt_wide=0, nt=0),
    "one-pass fwd, nn=c13 ntw=c13 nt=c13": setv(2, halves=1, nn=13, nt_wide=13, nt=13),
    "one-pass both, all c13": setv(2, halves=1, bhalves=1, nn=13, nt_wide=13, nt=13),
    "r5 tile walk: groups of 4 tile columns": (lambda: (setv(2)(), ops.set_option("gemm_group_n", 4))),
    "r5 tile walk: groups of 12 tile columns": (lambda: (setv(2)(), ops.set_option("gemm_group_n", 12))),
    "r5 tile walk: row-major (no groups)": (lambda: (setv(2)(), ops.set_option("gemm_group_n", 0))),
    "r5 grouped dW: 1 block per launch": setv(1),
    "r5 grouped dW: 3 blocks per launch": setv(3),
    # round 4: weight gradients per operator on the side lane, every tile its whole reduction (no slabs, no reduce), on the
    # 80 KB plain kernel — leaves LDS room for a forward / dX workgroup on the same CU, unlike the 160 KB grouped kernel
    "attention dK/dV: one workgroup per head (not the persistent 16-wave kernel)": (lambda: (setv(2)(), ops.set_option("attn_persistent", 0))),
    "attention dK/dV persistent only (attn_persistent 1: the round-3 default)": (lambda: (setv(2)(), ops.set_option("attn_persistent", 1))),
    "last block's weight gradients per operator (DW_PEROP_TAIL 1)": (lambda: (setv(2)(), setattr(fused, "DW_PEROP_TAIL", 1), setattr(fused, "DW_GROUP_TILES", 256))),
    "last two blocks' weight gradients per operator (DW_PEROP_TAIL 2)": (lambda: (setv(2)(), setattr(fused, "DW_PEROP_TAIL", 2), setattr(fused, "DW_GROUP_TILES", 256))),
    "default with DW_GROUP_TILES 256 (as shipped)": (lambda: (setv(2)(), setattr(fused, "DW_GROUP_TILES", 256))),
    "persistent attention on 192 workgroups": (lambda: (setv(2)(), ops.set_option("attn_pers_ctas", 192))),
    "persistent attention on 128 workgroups": (lambda: (setv(2)(), ops.set_option("attn_pers_ctas", 128))),
    "persistent attention on 224 workgroups": (lambda: (setv(2)(), ops.set_option("attn_pers_ctas", 224))),
    "persistent attention on 512 workgroups": (lambda: (setv(2)(), ops.set_option("attn_pers_ctas", 512))),
    "attention dQ AND dK/dV persistent (attn_persistent 3)": (lambda: (setv(2)(), ops.set_option("attn_persistent", 3))),
    "attention dQ persistent only (attn_persistent 2)": (lambda: (setv(2)(), ops.set_option("attn_persistent", 2))),
    "attention fwd + dQ + dK/dV persistent (attn_persistent 7)": (lambda: (setv(2)(), ops.set_option("attn_persistent", 7))),
    "attention fwd + dK/dV persistent (attn_persistent 5)": (lambda: (setv(2)(), ops.set_option("attn_persistent", 5))),
    "grouped ring variant 1 (4 slots = 128 KB: 32 KB of LDS left per CU)": setv(2, variant=1),
    "grouped ring variant 2 (5 slots, DMA 2 ahead)": setv(2, variant=2),
    "per-op dW, whole reduction, tn=c15 (192x128x64, 4 waves)": setv(0, tn=15, split=1),
    "per-op dW, whole reduction, tn=c14 (192x128x64, 8 waves)": setv(0, tn=14, split=1),
    "per-op dW, whole reduction, tn=c0 (128x128x64)": setv(0, tn=0, split=1),
    "per-op dW, split 2, tn=c15": setv(0, tn=15, split=2),
    "per-op dW, split 2, tn=c0": setv(0, tn=0, split=2),
    "per-op dW, r2 path (split-K heuristic, 128x128x32)": setv(0),
}
if len(sys.argv) > 2:
    sel = sys.argv[2:]
    VARIANTS = {k: v for k, v in VARIANTS.items() if any(s in k for s in sel)}


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step(img, labels)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for v in VARIANTS.values():
    v()
    run(3)
res = {k: [] for k in VARIANTS}
for rnd in range(5):
    for k, v in VARIANTS.items():
        v()
        res[k].append(run(10))
for k, v in res.items():
    print(f"{k:52s} median {statistics.median(v):7.3f} ms  min {min(v):7.3f}  all {[round(x, 2) for x in v]}")
