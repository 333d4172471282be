import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import cflearn_amd as C
DEV = "cuda"
cfg = dict(in_channels=3, out_channels=3, start_channels=320, num_heads=8, use_spatial_transformer=True,
           num_transformer_layers=1, num_res_blocks=2, attention_downsample_rates=(1, 2, 4),
           channel_multipliers=(1, 2, 4, 4), context_dim=None)
img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
m = C.build_module("unet_diffuser", config=cfg)
with torch.no_grad():
    for prm in m.parameters():
        if float(prm.abs().max()) == 0.0:
            prm.normal_(0.0, 0.02 if prm.dim() > 1 else 0.01)
g = torch.Generator().manual_seed(8765)
x = torch.randn(1, 3, img, img, generator=g).clamp_(-1, 1); t = torch.randint(0, 1000, (1,), generator=g); noise = torch.randn(1, 3, img, img, generator=g)
m = m.to(DEV); xd, td, nd = x.to(DEV), t.to(DEV), noise.to(DEV)
y = m(xd, timesteps=td, context=None); loss = torch.nn.functional.mse_loss(y.float(), nd); loss.backward(); torch.cuda.synchronize()
L0 = loss.item(); params = dict(m.named_parameters())
def loss_at():
    with torch.no_grad():
        return torch.nn.functional.mse_loss(m(xd, timesteps=td, context=None).float(), nd).item()
print("img", img, "loss", L0, "repeat", loss_at(), loss_at())
for k in ["input_blocks.10.0.conv1.weight", "residual.0.conv1.weight", "output_blocks.0.0.conv1.weight", "output_blocks.2.1.conv.weight", "input_blocks.7.1.blocks.0.attn1.out_linear.0.weight", "input_blocks.4.0.conv2.weight"]:
    prm = params[k]; gk = prm.grad.detach().clone(); gn = gk.norm().item()
    for frac in (0.03, 0.01, 0.003, 0.1):
        eps = frac * L0 / gn
        step = gk / gn * eps
        with torch.no_grad():
            prm.add_(step); lp = loss_at(); prm.sub_(2 * step); lm = loss_at(); prm.add_(step)
        print(f"{k:50s} pred dL/L {frac:5.3f} eps/|w| {eps / prm.detach().norm().item():.4f}  ratio {(lp - lm) / (2 * eps) / gn:.3f}   (L+ {lp:.6f} L- {lm:.6f})", flush=True)
