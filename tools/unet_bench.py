"""DDPM training step on the zoo `diffusion/ddpm` UNet (reference zoo/configs/diffusion/ddpm/default.json: start 320,
multipliers 1/2/4/4, 2 ResBlocks per level, SpatialTransformer at down-sampling rates 1/2/4, 8 heads) on one MI355X:
q_sample -> UNet fwd -> MSE -> bwd -> fused AdamW.  Synthetic images, random-init weights.  Prints one JSON line.

    python tools/unet_bench.py --img 256 --batch 1 --steps 3 --warmup 1
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import cflearn_amd as C  # noqa: E402
from cflearn_amd.diffusion import DDPMTrainStep, NoiseSchedule  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--img", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--start-channels", type=int, default=320)
    ap.add_argument("--in-channels", type=int, default=3)
    ap.add_argument("--context-dim", type=int, default=0, help="> 0: cross-attention on a [B, 77, dim] context")
    ap.add_argument("--no-side-stream", action="store_true", help="parameter gradients on the compute stream (A/B, race check)")
    ap.add_argument("--poison", action="store_true",
                    help="debug: every torch.empty() buffer (outputs, workspaces) is filled with NaN first, so a kernel "
                         "that reads memory it never wrote shows up as a NaN loss")
    args = ap.parse_args()
    if args.poison:
        _empty = torch.empty

        def _poisoned(*a, **k):
            t = _empty(*a, **k)
            if t.is_cuda and t.numel():
                t.fill_(float("nan")) if t.is_floating_point() else t.fill_(0x7F)
            return t

        torch.empty = _poisoned
    if args.no_side_stream:
        from cflearn_amd.functional import SideStream

        SideStream.enabled = False
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = dict(in_channels=args.in_channels, out_channels=args.in_channels, start_channels=args.start_channels,
               num_heads=8, use_spatial_transformer=True, num_transformer_layers=1, num_res_blocks=2,
               attention_downsample_rates=(1, 2, 4), channel_multipliers=(1, 2, 4, 4),
               context_dim=args.context_dim or None)
    m = C.build_module("unet_diffuser", config=cfg).to(dev)
    n_params = sum(p.numel() for p in m.parameters())
    ts = DDPMTrainStep(m, NoiseSchedule(device=dev), lr=1.0e-4)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(args.batch, args.in_channels, args.img, args.img, generator=g).to(dev)
    ctx = torch.randn(args.batch, 77, args.context_dim, generator=g).to(dev) if args.context_dim else None
    t = torch.randint(0, 1000, (args.batch,), generator=g).to(dev)
    eps = torch.randn(x.shape, generator=g).to(dev)
    first = None
    for i in range(args.warmup):
        loss = ts.step(x, ctx, timesteps=t, noise=eps)
        if i == 0:
            first = loss.item() / args.batch
            print(f"[unet_bench] first step done, loss {first:.4f}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = []
    for _ in range(args.steps):
        loss = ts.step(x, ctx, timesteps=t, noise=eps)
        losses.append(loss)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    last = loss.item() / args.batch
    print(json.dumps(dict(workload=f"DDPM UNet (zoo diffusion/ddpm) {args.img}^2, batch {args.batch}", params=n_params,
                          ms_per_step=round(dt * 1e3, 2), samples_per_s=round(args.batch / dt, 3),
                          first_loss=round(first, 5) if first is not None else None, last_loss=round(last, 5),
                          losses=[round(l.item() / args.batch, 7) for l in losses],
                          grad_checksum=float(ts.arena.flat_g.double().abs().sum().item()),
                          peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 2), steps=args.steps,
                          warmup=args.warmup, dtype="bf16", data="synthetic")))


if __name__ == "__main__":
    main()
