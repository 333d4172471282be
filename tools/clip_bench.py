"""CLIP contrastive training step (BASELINE config 5) on one MI355X: default `CLIP()` = ViT-B/32 image tower + 12-layer /
512-d / 8-head causal text tower (reference multimodal/clip.py:22-256), symmetric InfoNCE (contrastive.py; the reference
has no training loss: new design), backward, fused AdamW.  Synthetic batch per SURVEY §8d: images N(0,1), text
randint(1, 49407) with an EOT (49407) at a random position >= 8.  Prints one JSON line.

    python tools/clip_bench.py --batch 256 --steps 10 --warmup 3
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import cflearn_amd as C  # noqa: E402
from cflearn_amd.engine import LossTrainStep  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = C.build_module("clip", config={}).to(dev)
    n_params = sum(p.numel() for p in m.parameters())
    ts = LossTrainStep(m, lambda mod, b: mod.contrastive_loss(b["image"], b["text"]), lr=1.0e-4)
    g = torch.Generator().manual_seed(1234)
    img = torch.randn(args.batch, 3, 224, 224, generator=g)
    txt = torch.randint(1, 49407, (args.batch, 77), generator=g)
    eot = torch.randint(8, 77, (args.batch,), generator=g)
    for i in range(args.batch):
        txt[i, eot[i]] = 49407
        txt[i, eot[i] + 1:] = 0
    batch = dict(image=img.to(dev), text=txt.to(dev))
    first = None
    for i in range(args.warmup):
        loss = ts.step(batch)
        if i == 0:
            first = loss.item()
            print(f"[clip_bench] first step done, loss {first:.4f} (ln(B) = {torch.log(torch.tensor(float(args.batch))).item():.4f})",
                  file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = ts.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps(dict(workload=f"CLIP (ViT-B/32 + 12 x 512 text tower) contrastive step, batch {args.batch}",
                          params=n_params, ms_per_step=round(dt * 1e3, 2), samples_per_s=round(args.batch / dt, 1),
                          first_loss=None if first is None else round(first, 5), last_loss=round(loss.item(), 5),
                          peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 2), steps=args.steps,
                          warmup=args.warmup, dtype="bf16 (similarity / loss fp32)", data="synthetic")))


if __name__ == "__main__":
    main()
