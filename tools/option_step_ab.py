"""ViT-B/16 batch-128 training step with a library option off / on, alternating inside one process (3 rounds, 40 timed steps each),
shader clock and socket power of every timed window:
    python tools/option_step_ab.py attn_one_pass 0 1"""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402

import cflearn_amd as C  # noqa: E402
from cflearn_amd import fused, ops  # noqa: E402
from cflearn_amd.engine import TrainStep  # noqa: E402
from tools.gpu_telemetry import GpuTelemetry  # noqa: E402

name, values = sys.argv[1], [int(v) for v in sys.argv[2:]] or [0, 1]
dev = torch.device("cuda")
tel = GpuTelemetry(0).start()
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
ring = [(torch.randn(128, 3, 224, 224, generator=g).to(dev), torch.randint(0, 1000, (128,), generator=g).to(dev)) for _ in range(4)]
res = {v: [] for v in values}
for rep in range(3):
    for v in values:
        ops.set_option(name, v)
        fused._plans.clear()
        m = C.vit_b16_classifier(1000).to(dev)
        ts = TrainStep(m, lr=1e-4)
        for i in range(8):
            ts.step(*ring[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            loss = ts.step(*ring[i % 4])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s = tel.summary(t0, t1)
        res[v].append(round((t1 - t0) / 40 * 1e3, 3))
        print(f"rep {rep} {name}={v}: {(t1 - t0) / 40 * 1e3:.3f} ms/step  sclk {s['sclk_mhz_avg']} MHz  {s['power_w_avg']} W  loss {loss.item() / 128:.4f}", flush=True)
        del ts, m
        fused._plans.clear()
        torch.cuda.empty_cache()
print({f"{name}={v}": r for v, r in res.items()})
tel.stop()
