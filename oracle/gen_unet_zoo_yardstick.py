"""How far does the REFERENCE'S OWN bf16 execution of the full-size zoo UNet sit from its fp32 execution?  — TEST INFRASTRUCTURE ONLY.

Build container only (imports the reference from /root/reference through oracle/refharness).  Writes the small fixture
tests/golden/unet_zoo_yardstick.pt that `tests/test_gpu_unet.py::test_unet_zoo_full_size_step_vs_oracle` uses as its bound.

Why (round 5): the round-4 test measured its yardstick by running `oracle/unet_oracle.py` under `torch.autocast(bf16)`.  That
restatement is built from primitive ops (im2row + `@`, hand-written norms), and autocast only rounds its MATRIX PRODUCTS: the bias
add after a product is bf16 + f32 = f32, so the whole residual stream of that run stays in f32 — fewer roundings than the
reference's real mixed-precision run, where `F.conv2d` / `F.linear` return bf16 tensors and every residual add rounds (and fewer
than ours, which keeps the reference's bf16 stream).  The HIP path sat a systematic 1.03-1.28 x above that yardstick (VERDICT r4
weak #2).  The honest yardstick is the reference's own modules (cflearn.modules.multimodal.diffusion.unet.UNetDiffuser, the zoo
`diffusion/ddpm` configuration) under `torch.autocast("cpu", dtype=torch.bfloat16)` — what `mixed_precision="bf16"` executes
(trainer.py:264-273 via accelerate).  It cannot travel to the GPU box, so its NUMBERS do: per sampled tensor the rel-L2 distance
of the autocast gradient from the fp32 gradient, the same for the output, and the fp32 norms (so that the test can check that ITS
fp32 oracle reproduces the reference's fp32 gradients at full size — the restatement is pinned at 865 M parameters, not only
on the small fixtures).

    python oracle/gen_unet_zoo_yardstick.py        # ~10 min on 8 cores, 20 GB
"""
import importlib
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from refharness import load_reference  # noqa: E402

CFG = dict(in_channels=3, out_channels=3, start_channels=320, num_heads=8, use_spatial_transformer=True,
           num_transformer_layers=1, num_res_blocks=2, attention_downsample_rates=(1, 2, 4),
           channel_multipliers=(1, 2, 4, 4), context_dim=None)


def seeded_problem():
    """state dict + inputs exactly as tests/test_gpu_unet.py::test_unet_zoo_full_size_step_vs_oracle builds them (the module of this
    repo is only used for its seeded initialisation: parameter creation runs on the CPU, no kernel is involved)."""
    import cflearn_amd as C

    torch.manual_seed(0)
    m = C.build_module("unet_diffuser", config=CFG)
    with torch.no_grad():
        for prm in m.parameters():
            if float(prm.abs().max()) == 0.0:
                prm.normal_(0.0, 0.02 if prm.dim() > 1 else 0.01)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(1, 3, 64, 64, generator=g).clamp_(-1, 1)
    t = torch.randint(0, 1000, (1,), generator=g)
    noise = torch.randn(1, 3, 64, 64, generator=g)
    return sd, x, t, noise


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()


def main() -> None:
    from test_gpu_unet import ZOO_SAMPLED  # the tensors the GPU test samples (module import needs no GPU)

    load_reference()
    unet = importlib.import_module("cflearn.modules.multimodal.diffusion.unet")
    sd, x, t, noise = seeded_problem()
    m = unet.UNetDiffuser(**CFG)
    missing = m.load_state_dict(sd, strict=True)
    print("reference UNetDiffuser:", sum(p.numel() for p in m.parameters()), "parameters;", missing)
    params = dict(m.named_parameters())
    leaves = [params[k] for k in ZOO_SAMPLED]

    t0 = time.time()
    y32 = m(x, timesteps=t, context=None)
    loss32 = torch.nn.functional.mse_loss(y32, noise)
    g32 = torch.autograd.grad(loss32, leaves)
    print(f"fp32 forward + backward: {time.time() - t0:.1f} s, loss {loss32.item():.6f}", flush=True)

    t0 = time.time()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        y16 = m(x, timesteps=t, context=None)
        loss16 = torch.nn.functional.mse_loss(y16.float(), noise)
    g16 = torch.autograd.grad(loss16, leaves)
    print(f"bf16-autocast forward + backward: {time.time() - t0:.1f} s, loss {loss16.item():.6f}, output dtype {y16.dtype}", flush=True)

    out = dict(
        cfg=CFG, torch_version=torch.__version__, threads=torch.get_num_threads(),
        loss_fp32=loss32.item(), loss_autocast=loss16.item(),
        y_err=rel_l2(y16.detach(), y32.detach()), y_norm=y32.detach().norm().item(),
        grad_err={k: rel_l2(a, b) for k, a, b in zip(ZOO_SAMPLED, g16, g32)},
        grad_norm={k: b.norm().item() for k, b in zip(ZOO_SAMPLED, g32)},
        # a few fp32 values per tensor: the test compares its own oracle's fp32 gradients element-wise on them
        grad_probe={k: b.flatten()[:: max(1, b.numel() // 64)][:64].clone() for k, b in zip(ZOO_SAMPLED, g32)},
        y_probe=y32.detach().flatten()[::192][:64].clone(),
    )
    print(f"output: autocast vs fp32 rel-L2 {out['y_err']:.3e}")
    for k in ZOO_SAMPLED:
        print(f"    {k:55s} {out['grad_err'][k]:.3e}   |g| {out['grad_norm'][k]:.3e}")
    dst = os.path.join(ROOT, "tests", "golden", "unet_zoo_yardstick.pt")
    torch.save(out, dst)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
