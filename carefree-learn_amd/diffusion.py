"""DDPM training step around `modules.UNetDiffuser` (reference modules/multimodal/diffusion/ddpm.py:51-89,599-640;
samplers/schema.py:90-112; models/cv/diffusion.py:44-94): the float64 beta schedule (host, numpy, bit-exact with the
reference's arithmetic), the forward process `q_sample` and the epsilon-prediction MSE objective as kernels, and a
step engine on the flat arena + fused Adam like `engine.TrainStep`.

Scope: parameterization "eps", loss "l2", `log_var` fixed at its initial 0 (loss / exp(0) + 0), `l_simple_weight` 1,
`original_elbo_weight` 0 — the zoo `diffusion/ddpm` defaults."""
from typing import Any, Optional

import numpy as np
import torch
from torch import Tensor

from . import ops
from .functional import SideStream
from .optim import FusedAdam, ParamArena


class NoiseSchedule:
    """`make_beta_schedule` + `_register_noise_schedule` (ddpm.py:51-89,599-640): float64 on the host, fp32 buffers"""

    def __init__(self, timesteps: int = 1000, beta_schedule: str = "linear", linear_start: float = 8.5e-4,
                 linear_end: float = 1.2e-2, device: Any = "cpu"):
        if beta_schedule == "linear":
            betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
        elif beta_schedule == "sqrt_linear":
            betas = np.linspace(linear_start, linear_end, timesteps, dtype=np.float64)
        elif beta_schedule == "sqrt":
            betas = np.linspace(linear_start, linear_end, timesteps, dtype=np.float64) ** 0.5
        else:
            raise NotImplementedError(f"beta schedule '{beta_schedule}' is not built")
        alphas = 1.0 - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        to_t = lambda a: torch.from_numpy(a.astype(np.float32)).to(device)  # noqa: E731  (cftool.array.to_torch)
        self.t = timesteps
        self.betas = to_t(betas)
        self.alphas_cumprod = to_t(alphas_cumprod)
        self.sqrt_alphas_cumprod = to_t(np.sqrt(alphas_cumprod))
        self.sqrt_one_minus_alphas_cumprod = to_t(np.sqrt(1.0 - alphas_cumprod))

    def q_sample(self, x: Tensor, timesteps: Tensor, noise: Tensor, out_dtype: torch.dtype = torch.float32) -> Tensor:
        return ops.q_sample(x.float(), noise.float(), timesteps, self.sqrt_alphas_cumprod,
                            self.sqrt_one_minus_alphas_cumprod, out_dtype)


class DDPMTrainStep:
    """One optimisation step of the epsilon-prediction objective: x_t = q_sample(x, t, eps); eps_hat = unet(x_t, t,
    context); loss = mean_b mean_chw (eps_hat - eps)^2; backward; fused AdamW over the arena.  `t` and `eps` are drawn
    on the device when not given (torch's generator: input sampling, not part of the arithmetic path)."""

    def __init__(self, unet: torch.nn.Module, schedule: Optional[NoiseSchedule] = None, *, lr: float = 1.0e-4,
                 betas: Any = (0.9, 0.999), eps: float = 1.0e-8, weight_decay: float = 0.0, decoupled: bool = True,
                 distributed: bool = False, bucket_bytes: int = 256 << 20):
        self.unet = unet
        params = [p for p in unet.parameters() if p.requires_grad]
        dev = params[0].device
        self.schedule = schedule or NoiseSchedule(device=dev)
        self.arena = ParamArena(params, with_shadow=True)
        self.optimizer = FusedAdam(None, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled,
                                   arena=self.arena)
        self.optimizer.lazy_zero = True
        self.reducer = None
        if distributed:  # 3.46 GB of fp32 gradients per step: 256 MB buckets, overlapped with the backward
            from .ddp import BucketedAllReduce

            self.reducer = BucketedAllReduce(self.arena, bucket_bytes=bucket_bytes, optimizer=self.optimizer)
            self.reducer.broadcast_parameters(0)
        self.loss_sum: Optional[Tensor] = None

    def step(self, x: Tensor, context: Optional[Tensor] = None, *, timesteps: Optional[Tensor] = None,
             noise: Optional[Tensor] = None) -> Tensor:
        """Returns the device tensor holding sum_b of the per-sample MSE (divide by the batch for the mean loss)."""
        b = x.shape[0]
        if timesteps is None:
            timesteps = torch.randint(0, self.schedule.t, (b,), device=x.device, dtype=torch.int64)
        if noise is None:
            noise = torch.randn(x.shape, device=x.device, dtype=torch.float32)
        self.optimizer.prepare_step()
        self.optimizer.zero_grad()
        x_t = self.schedule.q_sample(x, timesteps, noise, torch.bfloat16)
        pred = self.unet(x_t, timesteps=timesteps, context=context)
        self.loss_sum, dpred = ops.mse_loss(pred, noise.float(), 1.0 / b)
        pred.backward(dpred)
        SideStream.join()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.launch_step()
        return self.loss_sum
