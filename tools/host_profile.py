"""Where the HOST spends a training step (cProfile over N eager steps of bench.py's UNet / CLIP / ViT workloads).
    python tools/host_profile.py unet|clip|vit [steps]      -> top functions by own time, per step"""
import cProfile
import os
import pstats
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "unet"
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 5
dev = torch.device("cuda")
torch.manual_seed(0)
import cflearn_amd as C  # noqa: E402

g = torch.Generator().manual_seed(1234)
if which == "unet":
    from cflearn_amd.diffusion import DDPMTrainStep, NoiseSchedule
    cfg = dict(in_channels=3, out_channels=3, start_channels=320, num_heads=8, use_spatial_transformer=True, num_transformer_layers=1,
               num_res_blocks=2, attention_downsample_rates=(1, 2, 4), channel_multipliers=(1, 2, 4, 4), context_dim=None)
    m = C.build_module("unet_diffuser", config=cfg).to(dev)
    ts = DDPMTrainStep(m, NoiseSchedule(device=dev), lr=1e-4)
    x = torch.randn(8, 3, 64, 64, generator=g).clamp_(-1, 1).to(dev)
    t = torch.randint(0, 1000, (8,), generator=g).to(dev)
    eps = torch.randn(x.shape, generator=g).to(dev)
    step = lambda: ts.step(x, None, timesteps=t, noise=eps)  # noqa: E731
elif which == "clip":
    from cflearn_amd.engine import LossTrainStep
    m = C.build_module("clip", config={}).to(dev)
    ts = LossTrainStep(m, lambda mod, b_: mod.contrastive_loss(b_["image"], b_["text"]), lr=1e-4)
    txt = torch.randint(1, 49407, (256, 77), generator=g)
    txt[:, 40] = 49407
    txt[:, 41:] = 0
    data = dict(image=torch.randn(256, 3, 224, 224, generator=g).to(dev), text=txt.to(dev))
    step = lambda: ts.step(data)  # noqa: E731
else:
    from cflearn_amd.engine import TrainStep
    m = C.vit_b16_classifier(1000).to(dev)
    ts = TrainStep(m, lr=1e-4)
    img = torch.randn(128, 3, 224, 224, generator=g).to(dev)
    lab = torch.randint(0, 1000, (128,), generator=g).to(dev)
    step = lambda: ts.step(img, lab)  # noqa: E731
if "--inline-backward" in sys.argv:  # run autograd's backward on this thread so that cProfile sees the Python backward functions
    torch.autograd.set_multithreading_enabled(False)
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
rows = []
for (fn, line, name), (cc, nc, tt, ct, _) in st.stats.items():
    rows.append((tt / steps * 1e3, ct / steps * 1e3, nc / steps, f"{os.path.basename(fn)}:{line} {name}"))
rows.sort(reverse=True)
print(f"{which}: host time per step by function (own ms, cumulative ms, calls) — profiler overhead included")
print(f"total own time {sum(r[0] for r in rows):.1f} ms/step")
for tt, ct, nc, name in rows[:60]:
    print(f"{tt:8.2f} {ct:8.2f} {nc:8.0f}  {name[:110]}")
