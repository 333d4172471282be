"""3x3 convolution forward (= the dX kernel with swapped channel counts) at the DDPM UNet's shapes, alone, per tile form.
    python tools/conv_bench.py [--forms=-1,0,1,2,3] [--splits] [--b256]
Every (pixels, Cin, Cout) the zoo UNet runs at 64^2 x 8 (forward and input gradient), with its count per step; forms and K splits
are forced through the `conv_form` / `conv_split` options (csrc/gemm.hip: conv_plan); every form's output is compared with the
default's (same K order inside a tile: bit-equal without a split).
(Round 3 also ran the phase kernel on four waves of 128x64 instead of eight of 128x32 — an experimental option that is
not in the tree: 1 752 -> 2 127 us over these shapes, profiles/r03/conv_bench_four_waves.txt.)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402

# (B, H, W, Cin, Cout, launches per step: forward + input gradient of the layers with this GEMM shape)
UNET64 = [
    (8, 64, 64, 320, 320, 14), (8, 64, 64, 640, 320, 2), (8, 64, 64, 320, 640, 2), (8, 64, 64, 960, 320, 1), (8, 64, 64, 320, 960, 1),
    (8, 64, 64, 640, 640, 2),
    (8, 32, 32, 320, 640, 1), (8, 32, 32, 640, 320, 1), (8, 32, 32, 640, 640, 12), (8, 32, 32, 1280, 1280, 2),
    (8, 32, 32, 1920, 640, 1), (8, 32, 32, 640, 1920, 1), (8, 32, 32, 1280, 640, 1), (8, 32, 32, 640, 1280, 1), (8, 32, 32, 960, 640, 1),
    (8, 32, 32, 640, 960, 1),
    (8, 16, 16, 640, 1280, 1), (8, 16, 16, 1280, 640, 1), (8, 16, 16, 1280, 1280, 14), (8, 16, 16, 2560, 1280, 2), (8, 16, 16, 1280, 2560, 2),
    (8, 16, 16, 1920, 1280, 1), (8, 16, 16, 1280, 1920, 1),
    (8, 8, 8, 1280, 1280, 22), (8, 8, 8, 2560, 1280, 3), (8, 8, 8, 1280, 2560, 3),
]
UNET256 = [(1, 256, 256, 320, 320, 14), (1, 256, 256, 640, 320, 2), (1, 256, 256, 320, 640, 2), (1, 128, 128, 640, 640, 12),
           (1, 64, 64, 1280, 1280, 14), (1, 32, 32, 1280, 1280, 22)]


def time_us(fn, iters):
    for _ in range(3):
        y = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y = fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--forms", default="-1,0,1,2,3")
    ap.add_argument("--splits", action="store_true", help="also sweep conv_split over 1, 2, 3, 4, 6, 8, 12, 16 per form")
    ap.add_argument("--b256", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    forms = [int(f) for f in a.forms.split(",")]
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    totals = {}
    best_total = 0.0
    for b, h, w, cin, cout, cnt in (UNET256 if a.b256 else UNET64):
        x = (torch.randn(b * h * w, cin, generator=g, device=dev) * 0.5).to(torch.bfloat16)
        wt = (torch.randn(cout, cin, 3, 3, generator=g, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(cout, generator=g, device=dev) * 0.1
        wk = ops.conv3x3_pack_filters(wt, False)
        fl = 2.0 * b * h * w * cout * 9 * cin
        ops.set_option("conv_form", -1)
        ops.set_option("conv_split", -1)
        ref = ops.conv3x3_nhwc(x, wk, bias, b, h, w).float()
        cells, best = [], (1e30, None)
        for form in forms:
            ops.set_option("conv_form", form)
            row_best = (1e30, None, 0.0)
            for split in ([-1, 1, 2, 3, 4, 6, 8, 12, 16] if a.splits else [-1]):
                ops.set_option("conv_split", split)
                us, y = time_us(lambda: ops.conv3x3_nhwc(x, wk, bias, b, h, w), a.iters)
                err = (y.float() - ref).abs().max().item()
                if us < row_best[0]:
                    row_best = (us, split, err)
            ops.set_option("conv_split", -1)
            totals[form] = totals.get(form, 0.0) + row_best[0] * cnt
            cells.append(f"f{form}: {row_best[0]:7.1f} us {fl / row_best[0] / 1e6:5.0f} TF" + (f" s{row_best[1]}" if a.splits else "") +
                         (f" !{row_best[2]:.1e}" if row_best[2] > 0 else ""))
            if row_best[0] < best[0]:
                best = (row_best[0], (form, row_best[1]))
        best_total += best[0] * cnt
        print(f"B{b} {h:>3}x{w:<3} {cin:>4}->{cout:<4} x{cnt:<2} | " + " | ".join(cells) + f" | best {best[1]}", flush=True)
    ops.set_option("conv_form", -1)
    print("per step (us x launches): " + ", ".join(f"form {f}: {t:.0f}" for f, t in totals.items()) + f"; best-of: {best_total:.0f}")


if __name__ == "__main__":
    main()
