"""How far does the REFERENCE'S OWN bf16 execution of ViT-B/16 (the headline configuration) sit from its fp32 execution, tensor by
tensor?  — TEST INFRASTRUCTURE ONLY.

Build container only (imports the reference from /root/reference through oracle/refharness).  Writes
tests/golden/vit_b16_yardstick.pt, the bound of `tests/test_gpu_modules.py::test_vit_b16_loss_within_1e3_of_cpu_reference` for its
152 parameter gradients (rounds 1-5 used one flat 2.5e-2; round 6's full-size CLIP test showed what a flat bound hides: the
one-word bf16 residual-gradient stream drifts where per-sample gradients cancel).  The question this fixture answers for the
headline model: is the one-word gradient stream (1.9 % faster than two words on the ViT-B/16 step, tools/grad_words_ab.py) within
1.1 x of the reference's own bf16 distance on EVERY tensor of the classifier?  If not, the classifier gets two words as well.

The model: cflearn's `ViTEncoder` (cv/encoder/transformer.py:17-100) + `Linear` head (`VanillaClassifier` =
head(encoder(x))), ViT-B/16 at 224^2, 1000 classes, the SAME seeded problem as the GPU test (this repo's module is used only for
its seeded CPU initialisation).  What travels: per tensor the rel-L2 distance of the reference's `torch.autocast("cpu", bfloat16)`
gradients from its fp32 gradients, the fp32 norms and 16 probe values (pins oracle/vit_oracle.py at full size), and the same for
the logits.

    python oracle/gen_vit_b16_yardstick.py        # ~1 min on 8 cores
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from refharness import load_reference  # noqa: E402

BATCH = 8


def seeded_problem(batch: int = BATCH):
    """state dict + inputs as the GPU test builds them (batch 8: enough samples that a batch-summed gradient means something)"""
    import cflearn_amd as C

    torch.manual_seed(0)
    m = C.vit_b16_classifier(num_classes=1000)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    img = torch.randn(batch, 3, 224, 224, generator=g)
    labels = torch.randint(0, 1000, (batch, 1), generator=g)
    return sd, img, labels


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()


def probe(t: torch.Tensor, n: int = 16) -> torch.Tensor:
    t = t.detach().flatten()
    return t[:: max(1, t.numel() // n)][:n].clone()


def main() -> None:
    ref = load_reference()
    sd, img, labels = seeded_problem()
    enc = ref.ViTEncoder(img_size=224, patch_size=16, in_channels=3, latent_dim=768, num_layers=12)
    head = ref.Linear(768, 1000)
    print("encoder:", enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, strict=True))
    print("head:", head.load_state_dict({k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}, strict=True))
    params = {f"encoder.{k}": p for k, p in enc.named_parameters()}
    params.update({f"head.{k}": p for k, p in head.named_parameters()})
    names = list(params)
    leaves = [params[k] for k in names]
    ce = torch.nn.functional.cross_entropy

    t0 = time.time()
    lg32 = head(enc(img))
    loss32 = ce(lg32, labels.view(-1))
    g32 = torch.autograd.grad(loss32, leaves)
    print(f"fp32 forward + backward: {time.time() - t0:.1f} s, loss {loss32.item():.6f}", flush=True)
    t0 = time.time()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lg16 = head(enc(img))
        loss16 = ce(lg16.float(), labels.view(-1))
    g16 = torch.autograd.grad(loss16, leaves)
    print(f"bf16-autocast forward + backward: {time.time() - t0:.1f} s, loss {loss16.item():.6f}, logits {lg16.dtype}", flush=True)

    out = dict(batch=BATCH, torch_version=torch.__version__, n_params=sum(p.numel() for p in leaves),
               loss_fp32=loss32.item(), loss_autocast=loss16.item(), logits_err=rel_l2(lg16.detach(), lg32.detach()),
               logits_probe=probe(lg32, 64),
               grad_err={k: rel_l2(a, b) for k, a, b in zip(names, g16, g32)},
               grad_norm={k: b.norm().item() for k, b in zip(names, g32)},
               grad_probe={k: probe(b) for k, b in zip(names, g32)})
    errs = sorted(out["grad_err"].values())
    print(f"logits: autocast vs fp32 {out['logits_err']:.3e}; {len(names)} gradients: min {errs[0]:.3e}, median {errs[len(errs) // 2]:.3e}, "
          f"max {errs[-1]:.3e}")
    for k in names:
        if ".mixing_blocks." not in k or ".mixing_blocks.0." in k or ".mixing_blocks.11." in k:
            print(f"    {k:70s} {out['grad_err'][k]:.3e}   |g| {out['grad_norm'][k]:.3e}")
    dst = os.path.join(ROOT, "tests", "golden", "vit_b16_yardstick.pt")
    torch.save(out, dst)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
