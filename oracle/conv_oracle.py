"""CPU restatement (fp32, plain torch tensor arithmetic) of the reference's conv / batch-norm / FCNN
building blocks — TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and
oracle/gen_golden.py; nothing under carefree-learn_amd/ may import it).

Each function cites the reference lines it restates (relative to /root/reference/cflearn/).  The
restatement is pinned: `oracle/gen_golden.py` checks every function against the reference's own
modules (imported through oracle/refharness) before it writes the fixtures under tests/golden/
(conv2d.pt, batchnorm.pt, mnist_clf.pt, fcnn.pt, layernorm4d.pt).
"""
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

StateDict = Dict[str, Tensor]


def bf16_round(t: Tensor) -> Tensor:
    """Round to bf16 and back, identity gradient: passed as `rnd` to the model functions below it places the
    roundings where the HIP path (and the reference under mixed_precision="bf16") stores bf16, so a test can
    separate "bf16 storage" error from kernel error.  Default `rnd` is the identity = the fp32 reference."""
    return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()


def _id(t: Tensor) -> Tensor:
    return t


def layer_norm_4d(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], eps: float = 1.0e-6) -> Tensor:
    """The reference's `LN.forward` on a 4-D input (modules/core/norms.py:30-46; built by NormFactory("layer_norm"), whose default
    config injects eps = 1e-6, norms.py:118-119): ONE mean and one UNBIASED standard deviation per sample over all C*H*W elements
    (torch.std's default Bessel correction), eps added to the standard deviation, then the per-channel affine.  Sums written out
    so that no torch reduction shortcut stands between the formula and the result."""
    b = x.shape[0]
    flat = x.reshape(b, -1)
    n = flat.shape[1]
    mean = flat.sum(1) / n
    centred = flat - mean[:, None]
    std = torch.sqrt((centred * centred).sum(1) / (n - 1))
    y = (x - mean.view(b, 1, 1, 1)) / (std.view(b, 1, 1, 1) + eps)
    if weight is not None:
        y = y * weight.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    return y


def conv2d(x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int = 1, padding: int = 0,
           dilation: int = 1) -> Tensor:
    """F.conv2d as called by Conv2d.forward with groups = 1 (modules/core/convs/basic.py:160-177):
    explicit im2row + matrix product, k ordered (c, ky, kx) like `weight.view(Cout, -1)`."""
    b, c, h, w = x.shape
    cout, _, kh, kw = weight.shape
    ho = (h + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    wo = (w + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    xp = torch.zeros((b, c, h + 2 * padding, w + 2 * padding), dtype=x.dtype)
    xp[:, :, padding:padding + h, padding:padding + w] = x
    cols = []
    for ky in range(kh):
        for kx in range(kw):
            y0, x0 = ky * dilation, kx * dilation
            cols.append(xp[:, :, y0:y0 + (ho - 1) * stride + 1:stride, x0:x0 + (wo - 1) * stride + 1:stride])
    rows = torch.stack(cols, dim=2)  # [B, C, kh*kw, Ho, Wo]
    rows = rows.permute(0, 3, 4, 1, 2).reshape(b * ho * wo, c * kh * kw)
    out = rows @ weight.reshape(cout, -1).t()
    if bias is not None:
        out = out + bias
    return out.view(b, ho, wo, cout).permute(0, 3, 1, 2)


def batch_norm_train(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float = 1.0e-5
                     ) -> Tuple[Tensor, Tensor, Tensor]:
    """nn.BatchNorm{1,2}d in training mode (modules/core/norms.py:20-27,90-93): per-channel statistics over
    every other dim, BIASED variance for the normalisation.  Returns (y, batch mean, biased batch var)."""
    dims = [d for d in range(x.dim()) if d != 1]
    shape = [1, -1] + [1] * (x.dim() - 2)
    mean = x.mean(dim=dims)
    var = ((x - mean.view(shape)) ** 2).mean(dim=dims)
    y = (x - mean.view(shape)) / torch.sqrt(var.view(shape) + eps)
    if gamma is not None:
        y = y * gamma.view(shape) + beta.view(shape)
    return y, mean, var


def running_update(running_mean: Tensor, running_var: Tensor, mean: Tensor, var_biased: Tensor, n: int,
                   momentum: float = 0.1) -> Tuple[Tensor, Tensor]:
    """running statistics after one training forward: the UNBIASED batch variance enters running_var"""
    unbiased = var_biased * n / max(n - 1, 1)
    return (1 - momentum) * running_mean + momentum * mean, (1 - momentum) * running_var + momentum * unbiased


def batch_norm_eval(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], running_mean: Tensor,
                    running_var: Tensor, eps: float = 1.0e-5) -> Tensor:
    shape = [1, -1] + [1] * (x.dim() - 2)
    y = (x - running_mean.view(shape)) / torch.sqrt(running_var.view(shape) + eps)
    if gamma is not None:
        y = y * gamma.view(shape) + beta.view(shape)
    return y


def leaky_relu(x: Tensor, slope: float) -> Tensor:
    """nn.LeakyReLU(slope) / nn.ReLU for slope 0 (modules/core/activations.py:35-48)"""
    return torch.where(x > 0, x, x * slope)


def vanilla_encoder_1d(x: Tensor, sd: StateDict, prefix: str, num_downsample: int, slope: float = 0.2,
                       first_kernel_size: int = 7, kernel_size: int = 3, eps: float = 1.0e-5,
                       training: bool = True, rnd=_id) -> Tensor:
    """VanillaEncoder1D (modules/cv/encoder/vanilla.py:18-158): conv k7 s1 'same' -> BN -> LeakyReLU(0.2), then
    `num_downsample` x [conv k3 s2 pad 1 (-> BN -> LeakyReLU except after the last)], AdaptiveAvgPool2d((1,1)),
    squeeze.  `prefix` addresses the nn.Sequential (e.g. 'encoder.encoder.encoder.').  Training-mode BN."""
    def bn(net: Tensor, i: int) -> Tensor:
        if training:
            return batch_norm_train(net, sd[f"{prefix}{i}.weight"], sd[f"{prefix}{i}.bias"], eps)[0]
        return batch_norm_eval(net, sd[f"{prefix}{i}.weight"], sd[f"{prefix}{i}.bias"],
                               sd[f"{prefix}{i}.running_mean"], sd[f"{prefix}{i}.running_var"], eps)

    net = rnd(conv2d(rnd(x), rnd(sd[f"{prefix}0.weight"]), sd[f"{prefix}0.bias"], 1, first_kernel_size // 2))
    net = rnd(leaky_relu(rnd(bn(net, 1)), slope))
    idx = 3
    for i in range(num_downsample):
        net = rnd(conv2d(net, rnd(sd[f"{prefix}{idx}.weight"]), sd[f"{prefix}{idx}.bias"], 2, kernel_size // 2))
        if i != num_downsample - 1:
            net = rnd(leaky_relu(rnd(bn(net, idx + 1)), slope))
            idx += 3
    return rnd(net.mean(dim=(2, 3)))


def mnist_classifier(x: Tensor, sd: StateDict, num_downsample: int, training: bool = True, rnd=_id) -> Tensor:
    """VanillaClassifier(encoder='vanilla_1d') (modules/cv/classifier/vanilla.py:16-66): encoder + Linear head"""
    latent = vanilla_encoder_1d(x, sd, "encoder.encoder.encoder.", num_downsample, training=training, rnd=rnd)
    return latent @ rnd(sd["head.linear.weight"]).t() + sd["head.linear.bias"]


def fcnn(x: Tensor, sd: StateDict, num_hidden: int, rnd=_id) -> Tensor:
    """FCNN with the default Mapping blocks (modules/ml/fcnn.py:12-57; mappings.py:79-87 with batch_norm False,
    dropout 0): Linear -> ReLU per hidden unit, then nn.Linear"""
    net = rnd(x)
    for i in range(num_hidden):
        net = rnd(net @ rnd(sd[f"net.{i}.linear.linear.weight"]).t() + sd[f"net.{i}.linear.linear.bias"])
        net = rnd(leaky_relu(net, 0.0))
    return net @ rnd(sd[f"net.{num_hidden}.weight"]).t() + sd[f"net.{num_hidden}.bias"]
