"""What do the GELU / GELU' epilogues cost beyond their second operand?  The FF1 (+GELU) and FF2-dX (x GELU') GEMMs of ViT-B/16 at the
half-batch shape, timed hot (>= 1 s each, telemetry of the second half) with the product library and with a probe build whose
`gelu_parts` does no arithmetic (tools/build_variant.sh geluprobe -DCFHIP_GELU_PROBE; select with CFHIP_LIB):
    python tools/gelu_probe.py; CFHIP_LIB=tools/libcfhip_geluprobe.so python tools/gelu_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402
from tools.energy_table import loop  # noqa: E402
from tools.gpu_telemetry import GpuTelemetry  # noqa: E402

dev = torch.device("cuda")
tel = GpuTelemetry(0).start()
g = torch.Generator(device=dev).manual_seed(0)
bf = torch.bfloat16
rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).to(bf)  # noqa: E731
m, d, ff = 12608, 768, 3072
x, w1, b1 = rnd(m, d), rnd(ff, d), torch.randn(ff, device=dev, generator=g)
pre, h = torch.empty(m, ff, dtype=bf, device=dev), torch.empty(m, ff, dtype=bf, device=dev)
dy, w2 = rnd(m, d), rnd(d, ff)
dpre = torch.empty(m, ff, dtype=bf, device=dev)
rows = [
    ("FF1 + bias", lambda: ops.gemm(x, w1, bias=b1, out=h)),
    ("FF1 + bias + GELU (writes pre and h)", lambda: ops.gemm(x, w1, bias=b1, epilogue=ops.EPI_GELU, aux_out=pre, out=h)),
    ("FF2 dX plain (nn 12608x3072x768)", lambda: ops.gemm(dy, w2, b_trans=True, out=dpre)),
    ("FF2 dX x GELU'(pre)", lambda: ops.gemm(dy, w2, b_trans=True, epilogue=ops.EPI_DGELU, aux_in=pre, out=dpre)),
]
print("library:", os.environ.get("CFHIP_LIB", "product"))
time.sleep(1.0)
for name, fn in rows:
    n, sec, s = loop(fn, 1.5, tel)
    print(f"  {name:42s} {sec * 1e6:7.1f} us  {2.0 * m * ff * d / sec / 1e12:7.1f} TFLOP/s  {s['power_w_avg']} W  {s['sclk_mhz_avg']} MHz", flush=True)
tel.stop()
