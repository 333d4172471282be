"""Forward / backward phases of the zoo-UNet DDPM step: what the HOST needs to issue each and what the GPU needs to run it.
    python tools/unet_phases.py [img] [batch] [steps]     (CFHIP_TAPED_NODES=0/1, CFHIP_UNET_NHWC=0/1 as in bench.py)
Per step: host time inside UNetDiffuser.forward, host time for the rest (loss, backward, join, optimizer), the GPU's span between the
events recorded at the same three points, and the number of autograd nodes behind the prediction."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cflearn_amd as C  # noqa: E402
from cflearn_amd.diffusion import DDPMTrainStep, NoiseSchedule  # noqa: E402

img = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda")
cfg = dict(in_channels=3, out_channels=3, start_channels=320, num_heads=8, use_spatial_transformer=True, num_transformer_layers=1,
           num_res_blocks=2, attention_downsample_rates=(1, 2, 4), channel_multipliers=(1, 2, 4, 4), context_dim=None)
torch.manual_seed(0)
m = C.build_module("unet_diffuser", config=cfg).to(dev)
ts = DDPMTrainStep(m, NoiseSchedule(device=dev), lr=1e-4)
g = torch.Generator().manual_seed(1234)
x = torch.randn(batch, 3, img, img, generator=g).clamp_(-1, 1).to(dev)
t = torch.randint(0, 1000, (batch,), generator=g).to(dev)
eps = torch.randn(x.shape, generator=g).to(dev)
marks = {}


def pre(mod, args):
    marks["t0"] = time.perf_counter()
    marks["e0"] = torch.cuda.Event(enable_timing=True)
    marks["e0"].record()


def post(mod, args, out):
    marks["t1"] = time.perf_counter()
    marks["e1"] = torch.cuda.Event(enable_timing=True)
    marks["e1"].record()
    seen, stack = set(), [out.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        stack.extend(nf for nf, _ in fn.next_functions)
    marks["nodes"] = len(seen)
    marks["taped"] = sum(1 for fn in seen if type(fn).__name__ == "TapedFnBackward")


m.register_forward_pre_hook(pre)
m.register_forward_hook(post)
rows = []
for i in range(steps + 3):
    torch.cuda.synchronize()
    ts.step(x, None, timesteps=t, noise=eps)
    t2 = time.perf_counter()
    e2 = torch.cuda.Event(enable_timing=True)
    e2.record()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if i >= 3:
        rows.append(((marks["t1"] - marks["t0"]) * 1e3, (t2 - marks["t1"]) * 1e3, marks["e0"].elapsed_time(marks["e1"]),
                     marks["e1"].elapsed_time(e2), (t3 - marks["t0"]) * 1e3))
med = [sorted(r[k] for r in rows)[len(rows) // 2] for k in range(5)]
print(f"img {img} batch {batch} taped={os.environ.get('CFHIP_TAPED_NODES', '1')} nhwc={os.environ.get('CFHIP_UNET_NHWC', '1')}: "
      f"autograd nodes {marks['nodes']} ({marks['taped']} taped) | host forward {med[0]:.2f} ms, host backward+step {med[1]:.2f} | "
      f"GPU forward span {med[2]:.2f}, GPU backward+step span {med[3]:.2f} | wall {med[4]:.2f}")
