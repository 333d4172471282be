"""3x3 convolution forward (= the dX kernel) at the DDPM UNet's shapes, alone.   python tools/conv_bench.py
(Round 3 also ran it with the phase kernel on four waves of 128x64 instead of eight of 128x32 — an experimental option that is
not in the tree: 1 752 -> 2 127 us over these shapes, profiles/r03/conv_bench_four_waves.txt.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402

SHAPES = [(8, 64, 64, 320, 320), (8, 64, 64, 640, 320), (8, 64, 64, 960, 320), (8, 32, 32, 640, 640), (8, 32, 32, 1280, 640),
          (8, 32, 32, 320, 640), (8, 16, 16, 1280, 1280), (8, 16, 16, 2560, 1280), (8, 16, 16, 640, 1280), (8, 8, 8, 1280, 1280),
          (8, 8, 8, 2560, 1280), (1, 256, 256, 320, 320), (1, 128, 128, 640, 640), (1, 64, 64, 1280, 1280)]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
tot = {0: 0.0, 1: 0.0}
for b, h, w, cin, cout in SHAPES:
    x = (torch.randn(b * h * w, cin, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    wt = (torch.randn(cout, cin, 3, 3, generator=g, device=dev) * 0.05).to(torch.bfloat16)
    wk = ops.conv3x3_pack_filters(wt, False)
    res, outs = {}, {}
    for four in (0,):
        for _ in range(3):
            y = ops.conv3x3_nhwc(x, wk, None, b, h, w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = ops.conv3x3_nhwc(x, wk, None, b, h, w)
        e1.record()
        e1.synchronize()
        res[four] = e0.elapsed_time(e1) * 1e3 / 20
        outs[four] = y
        tot[four] += res[four]
    fl = 2.0 * b * h * w * cout * 9 * cin
    print(f"B{b} {h}x{w} Cin {cin:>4} Cout {cout:>4} | {res[0]:8.1f} us {fl / res[0] / 1e6:6.0f} TFLOP/s")
print(f"sum: {tot[0]:.0f} us")
