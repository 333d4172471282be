"""DDPM training step around `modules.UNetDiffuser` (reference modules/multimodal/diffusion/ddpm.py:51-89,599-640;
samplers/schema.py:90-112; models/cv/diffusion.py:44-94): the float64 beta schedule (host, numpy, bit-exact with the
reference's arithmetic), the forward process `q_sample` and the epsilon-prediction MSE objective as kernels, and a
step engine on the flat arena + fused Adam like `engine.TrainStep`.

Every objective of `DDPMStep.loss_fn` is built: parameterization "eps" / "x0" / "v", loss "l2" / "l1", a fixed or learned
per-timestep `log_var`, `l_simple_weight` and the `original_elbo_weight * lvlb_weights[t]` term; the zoo `diffusion/ddpm`
defaults are eps / l2 / log_var 0 / weights 1 and 0."""
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
                 linear_end: float = 1.2e-2, device: Any = "cpu", *, cosine_s: float = 8.0e-3,
                 given_betas: Optional[np.ndarray] = None, parameterization: str = "eps", v_posterior: float = 0.0):
        if given_betas is not None:
            betas = np.asarray(given_betas, dtype=np.float64)
            timesteps = len(betas)
        elif beta_schedule == "linear":
            betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
        elif beta_schedule == "cosine":
            steps = np.arange(timesteps + 1, dtype=np.float64) / timesteps + cosine_s
            alphas_c = np.cos(steps / (1 + cosine_s) * np.pi / 2) ** 2
            alphas_c = alphas_c / alphas_c[0]
            betas = np.clip(1.0 - alphas_c[1:] / alphas_c[:-1], 0.0, 0.999)
        elif beta_schedule == "sqrt_linear":
            betas = np.linspace(linear_start, linear_end, timesteps, dtype=np.float64)
        elif beta_schedule == "sqrt":
            betas = np.linspace(linear_start, linear_end, timesteps, dtype=np.float64) ** 0.5
        else:
            raise ValueError(f"unrecognized schedule '{beta_schedule}' occurred")
        if parameterization not in ("eps", "x0", "v"):
            raise NotImplementedError(f"unrecognized parameterization '{parameterization}' occurred")
        alphas = 1.0 - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1.0, alphas_cumprod[:-1])
        one_m_cumprod = 1.0 - alphas_cumprod
        to_t = lambda a: torch.from_numpy(a.astype(np.float32)).to(device)  # noqa: E731  (cftool.array.to_torch)
        self.t = timesteps
        self.parameterization = parameterization
        self.betas = to_t(betas)
        self.alphas_cumprod = to_t(alphas_cumprod)
        self.sqrt_alphas_cumprod = to_t(np.sqrt(alphas_cumprod))
        self.sqrt_one_minus_alphas_cumprod = to_t(np.sqrt(one_m_cumprod))
        self.neg_sqrt_one_minus_alphas_cumprod = -self.sqrt_one_minus_alphas_cumprod
        # q(x_{t-1} | x_t, x_0) and the variational-bound weights (ddpm.py:636-679): the reference forms them from the
        # fp32 buffers with torch arithmetic — same operations, same order, here
        posterior_variance = to_t(v_posterior * betas + (1.0 - v_posterior) * betas * (1.0 - alphas_cumprod_prev) / one_m_cumprod)
        self.posterior_variance = posterior_variance
        if parameterization == "eps":
            lvlb = 0.5 * self.betas ** 2 / (posterior_variance * to_t(alphas) * (1.0 - self.alphas_cumprod))
        elif parameterization == "x0":
            lvlb = to_t(0.25 * np.sqrt(alphas_cumprod) / one_m_cumprod)
        else:
            lvlb = torch.ones_like(self.betas ** 2 / (2 * posterior_variance * to_t(alphas) * (1.0 - self.alphas_cumprod)))
        lvlb[0] = lvlb[1]
        self.lvlb_weights = lvlb

    def q_sample(self, x: Tensor, timesteps: Tensor, noise: Tensor, out_dtype: torch.dtype = torch.float32) -> Tensor:
        return ops.q_sample(x.float(), noise.float(), timesteps, self.sqrt_alphas_cumprod,
                            self.sqrt_one_minus_alphas_cumprod, out_dtype)

    def v_target(self, x: Tensor, timesteps: Tensor, noise: Tensor) -> Tensor:
        """`DDPMStep.get_v` (models/cv/diffusion.py:96-101): sqrt(ac[t]) noise - sqrt(1 - ac[t]) x, the same kernel as
        q_sample with the operands' roles exchanged"""
        return ops.q_sample(noise.float(), x.float(), timesteps, self.sqrt_alphas_cumprod,
                            self.neg_sqrt_one_minus_alphas_cumprod, torch.float32)


class DDPMTrainStep:
    """One optimisation step of `DDPMStep.loss_fn` (models/cv/diffusion.py:44-94): x_t = q_sample(x, t, noise);
    out = unet(x_t, t, context); target = noise | x | v by `parameterization`; per sample loss_b = mean_chw f(out - target)
    (f = square | abs); loss = l_simple_weight * mean_b(loss_b / exp(log_var[t_b]) + log_var[t_b]) +
    original_elbo_weight * mean_b(lvlb_weights[t_b] * loss_b); backward; fused AdamW over the arena.  `log_var` is a
    per-timestep table (`log_var_init`), a trained parameter of the arena with `learn_log_var`.  `t` and `noise` are
    drawn on the device when not given (torch's generator: input sampling, not part of the arithmetic path)."""

    def __init__(self, unet: torch.nn.Module, schedule: Optional[NoiseSchedule] = None, *, lr: float = 1.0e-4,
                 betas: Any = (0.9, 0.999), eps: float = 1.0e-8, weight_decay: float = 0.0, decoupled: bool = True,
                 distributed: bool = False, bucket_bytes: int = 256 << 20, loss_type: str = "l2",
                 l_simple_weight: float = 1.0, original_elbo_weight: float = 0.0, learn_log_var: bool = False,
                 log_var_init: float = 0.0, use_graph: bool = False, step_in_backward: bool = True,
                 range_bytes: int = 64 << 20):
        if loss_type not in ("l1", "l2"):
            raise ValueError(f"unrecognized loss '{loss_type}' occurred")
        self.unet = unet
        params = [p for p in unet.parameters() if p.requires_grad]
        dev = params[0].device
        self.schedule = schedule or NoiseSchedule(device=dev)
        self.loss_type, self.l_simple_weight = loss_type, float(l_simple_weight)
        self.original_elbo_weight = float(original_elbo_weight)
        self.learn_log_var = learn_log_var
        self.log_var = torch.full((self.schedule.t,), float(log_var_init), dtype=torch.float32, device=dev)
        if learn_log_var:  # ddpm.py:231-236: an nn.Parameter trained with the UNet
            self.log_var = torch.nn.Parameter(self.log_var, requires_grad=True)
            params = params + [self.log_var]
        self.arena = ParamArena(params, with_shadow=True)
        self.optimizer = FusedAdam(None, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled,
                                   arena=self.arena)
        self.optimizer.lazy_zero = True
        self.reducer = None
        # the optimizer update inside backward (optim.StepInBackward): 865 M parameters x 30 bytes is 4.5 ms of HBM traffic that
        # used to run alone at the end of a 62 ms step
        in_bwd = bool(step_in_backward) and not use_graph and dev.type == "cuda"
        if distributed:  # 3.46 GB of fp32 gradients per step: 256 MB buckets, overlapped with the backward
            from .ddp import BucketedAllReduce

            self.reducer = BucketedAllReduce(self.arena, bucket_bytes=bucket_bytes, optimizer=self.optimizer, step_in_backward=in_bwd)
            self.reducer.broadcast_parameters(0)
        elif in_bwd:
            self.optimizer.enable_step_in_backward(range_bytes)
        self.loss_sum: Optional[Tensor] = None
        self.losses: dict = {}
        self.use_graph = bool(use_graph) and not distributed
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._static: dict = {}
        self._graph_loss: Optional[Tensor] = None

    def target(self, x: Tensor, timesteps: Tensor, noise: Tensor) -> Tensor:
        p = self.schedule.parameterization
        if p == "eps":
            return noise.float()
        if p == "x0":
            return x.float()
        return self.schedule.v_target(x, timesteps, noise)

    def objective(self, pred: Tensor, target: Tensor, timesteps: Tensor, want_grad: bool = True):
        """Returns (loss f32 scalar tensor, dpred bf16, dict of the reference's loss entries as device tensors);
        writes the gradient of a learned `log_var`."""
        from .functional import write_param_grad

        b = pred.shape[0]
        log_var_t = self.log_var.detach()[timesteps]
        inv = torch.exp(-log_var_t)
        weight = self.l_simple_weight * inv / b
        if self.original_elbo_weight > 0:
            weight = weight + self.original_elbo_weight * self.schedule.lvlb_weights[timesteps] / b
        per_sample, dpred = ops.diffusion_loss(pred, target, weight.float().contiguous(), self.loss_type, want_grad)
        loss_simple = per_sample / torch.exp(log_var_t) + log_var_t
        losses = {"simple": per_sample.mean()}
        if self.learn_log_var:
            losses["gamma"] = loss_simple.mean()
            losses["log_var"] = self.log_var.detach().mean()
            if want_grad:  # d/d log_var[t_b] of l_simple_weight * mean_b(loss_b exp(-lv) + lv)
                g = self.l_simple_weight * (1.0 - per_sample * inv) / b

                def scatter(out: Tensor, acc: bool) -> None:
                    if not acc:
                        out.zero_()
                    out.index_add_(0, timesteps, g)

                write_param_grad(self.log_var, scatter)
        loss = self.l_simple_weight * loss_simple.mean()
        if self.original_elbo_weight > 0:
            vlb = (self.schedule.lvlb_weights[timesteps] * per_sample).mean()
            losses["vlb"] = vlb
            loss = loss + self.original_elbo_weight * vlb
        losses["loss"] = loss
        return loss, dpred, losses

    def step(self, x: Tensor, context: Optional[Tensor] = None, *, timesteps: Optional[Tensor] = None,
             noise: Optional[Tensor] = None, labels: Optional[Tensor] = None) -> Tensor:
        """Returns the device tensor holding the loss of this step (`self.losses`: the reference's other entries)."""
        b = x.shape[0]
        if timesteps is None:
            timesteps = torch.randint(0, self.schedule.t, (b,), device=x.device, dtype=torch.int64)
        if noise is None:
            noise = torch.randn(x.shape, device=x.device, dtype=torch.float32)
        if self.use_graph:
            return self._graph_step(x, context, timesteps, noise, labels)
        self.optimizer.prepare_step()
        return self._body(x, context, timesteps, noise, labels)

    def _body(self, x: Tensor, context: Optional[Tensor], timesteps: Tensor, noise: Tensor, labels: Optional[Tensor]) -> Tensor:
        """All device work of one step, no host synchronisation (what a hipGraph records)."""
        b = x.shape[0]
        self.optimizer.zero_grad()
        x_t = self.schedule.q_sample(x, timesteps, noise, torch.bfloat16)
        kw = {} if labels is None else {"labels": labels}
        pred = self.unet(x_t, timesteps=timesteps, context=context, **kw)
        loss, dpred, self.losses = self.objective(pred, self.target(x, timesteps, noise), timesteps)
        self.loss_sum = loss * b  # (round-1 name: sum over the batch of the per-sample objective)
        pred.backward(dpred)
        SideStream.join()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.launch_step()
        return loss

    def _graph_step(self, x: Tensor, context: Optional[Tensor], timesteps: Tensor, noise: Tensor,
                    labels: Optional[Tensor]) -> Tensor:
        """`use_graph`: the step is recorded ONCE into a hipGraph (after two eager passes for the allocator, the side streams
        and the lazy initialisations, whose updates are rolled back) and replayed; the ~2 900 launches of a UNet step then cost the host one call (the eager step at
        64^2 x 8 is bound by the host's issue rate: 65 ms of Python per 69 ms step).  Inputs are copied into the recorded
        buffers; shapes (and the presence of context / labels) must not change between steps."""
        ins = dict(x=x, context=context, timesteps=timesteps, noise=noise, labels=labels)
        if self._graph is None:
            self._static = {k: (None if v is None else v.clone()) for k, v in ins.items()}
            snap = self.optimizer.snapshot()  # the two warm-up passes below are not training steps
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.optimizer.prepare_step()
                    self._body(**self._static)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.optimizer.restore(snap)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):  # records, does not execute: this call goes on to replay it once
                self._graph_loss = self._body(**self._static)
        for k, v in ins.items():
            st = self._static[k]
            if (v is None) != (st is None) or (v is not None and v.shape != st.shape):
                raise ValueError(f"DDPMTrainStep(use_graph=True): '{k}' changed its shape / presence after the recording")
            if v is not None and v.data_ptr() != st.data_ptr():
                st.copy_(v, non_blocking=True)
        self.optimizer.prepare_step()
        self._graph.replay()
        return self._graph_loss.clone()  # (the recorded tensor is overwritten by every replay)
