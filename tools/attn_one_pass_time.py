"""ViT-B/16 attention backward (B x 12 heads x 197 x 64): the two-pass kernels (dQ pass + dK / dV pass, persistent forms) against the
one-pass kernel of round 6 (option "attn_one_pass"), hot (>= 1.5 s each) with power / clock of the second half of each window.
    python tools/attn_one_pass_time.py [batch=64]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cflearn_amd import ops  # noqa: E402
from tools.energy_table import loop  # noqa: E402
from tools.gpu_telemetry import GpuTelemetry  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, H, D = 197, 12, 768
dev = "cuda"
tel = GpuTelemetry(0).start()
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B, T, 3 * D, device=dev, generator=g).to(torch.bfloat16)
d_o = torch.randn(B, T, D, device=dev, generator=g).to(torch.bfloat16)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
o, lse = ops.attn_fwd(q, k, v, H)
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
delta = torch.empty(B, H, T, device=dev)
flops = 3.5 * 4.0 * B * H * T * T * 64
time.sleep(1.0)
t0 = time.perf_counter()
time.sleep(2.0)
idle = tel.summary(t0, time.perf_counter())["power_w_avg"]
print(f"batch {B}: {B * H} heads; idle {idle} W")
for rnd in range(2):
    for one in (0, 1, 2):
        ops.set_option("attn_one_pass", one)
        n, sec, s = loop(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, H, dq=dq, dk=dk, dv=dv, parts=3, delta=delta), 1.5, tel)
        print(f"  {('two passes', 'one pass (16 waves)', 'one pass (8 waves, 2 key tiles)')[one]:31s}: {sec * 1e6:7.1f} us per backward  {flops / sec / 1e12:6.1f} TFLOP/s (two-pass FLOP count)  "
              f"{s['power_w_avg']} W  {s['sclk_mhz_avg']} MHz  {(s['power_w_avg'] - idle) * sec * 1e3:7.2f} mJ", flush=True)
ops.set_option("attn_one_pass", 2)
tel.stop()
