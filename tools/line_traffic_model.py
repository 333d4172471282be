"""Whole-line L2 -> LDS traffic of the GEMM tile configurations: bytes of 128-byte lines a workgroup pulls per
MFLOP, per layout.  A k-major operand (row = K contiguous) staged BK k per K-step touches ceil(2 BK / 128) lines per row and
uses 2 BK bytes of them; an m-major operand (row of the tile = one k) is contiguous along the tile.  No GPU needed.

    python tools/line_traffic_model.py
"""
CONFIGS = [("128x128x32 (c1)", 128, 128, 32), ("128x128x64 (c0)", 128, 128, 64), ("192x128x64 (c14)", 192, 128, 64),
           ("256x128x32 (c8)", 256, 128, 32), ("256x128x64", 256, 128, 64), ("256x256x32 (c7)", 256, 256, 32),
           ("256x256x64 (c13)", 256, 256, 64)]
LINE = 128


def operand_bytes(rows: int, bk: int, k_major: bool) -> int:
    if k_major:  # `rows` rows of 2 * bk bytes each, every row in its own line(s)
        return rows * -(-2 * bk // LINE) * LINE
    return bk * -(-2 * rows // LINE) * LINE  # bk k-rows of 2 * rows bytes


print(f"{'configuration':20s} {'nt (fwd)':>10s} {'nn (dX)':>10s} {'tn (dW)':>10s}   KB of lines per MFLOP (useful bytes in brackets)")
for name, bm, bn, bk in CONFIGS:
    mflop = 2.0 * bm * bn * bk / 1e6
    useful = (bm + bn) * bk * 2 / 1024 / mflop
    cells = []
    for a_kmajor, b_kmajor in ((True, True), (True, False), (False, False)):
        kb = (operand_bytes(bm, bk, a_kmajor) + operand_bytes(bn, bk, b_kmajor)) / 1024
        cells.append(f"{kb / mflop:10.1f}")
    print(f"{name:20s} {''.join(cells)}   ({useful:.1f})")
print("\nMeasured alone, 25216 x 768 x 3072 (profiles/r02/gemm_bench_b128_bk64.log): nt c0 124 us / c14 129 / c8 150 / c1 153;"
      "\nnn c14 129 / c8 144 / c0 152 / c1 158; tn (768 x 3072 x 25216) c14 138-153 / c1 160-177 / c8 164 / c0 217.")
