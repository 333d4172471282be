"""CPU restatement (fp32) of the UNet residual-block pieces — TEST INFRASTRUCTURE ONLY.

Follows cflearn/modules/core/convs/residual.py:86-253 (`ResDownsample`, `ResUpsample`,
`ResidualBlockWithTimeEmbedding._forward`) and cflearn/modules/multimodal/diffusion/unet.py:52-74
(`timestep_embedding`).  Pinned by oracle/gen_golden.py::gen_resblock against the reference classes imported through
oracle/refharness.  `safe_clip_` (toolkit.py:1236) only touches non-finite values and is left out.
"""
import math
from typing import Dict, Optional

import torch
from torch import Tensor

import conv_oracle as CO

StateDict = Dict[str, Tensor]


def group_norm(x: Tensor, gamma: Tensor, beta: Tensor, groups: int = 32, eps: float = 1.0e-6) -> Tensor:
    """nn.GroupNorm: statistics per (sample, group) over (C/G, H, W), biased variance"""
    b, c = x.shape[:2]
    xg = x.reshape(b, groups, -1)
    mean = xg.mean(dim=2, keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=2, keepdim=True)
    y = ((xg - mean) / torch.sqrt(var + eps)).reshape(x.shape)
    shape = [1, c] + [1] * (x.dim() - 2)
    return y * gamma.view(shape) + beta.view(shape)


def silu(x: Tensor) -> Tensor:
    return x / (1.0 + torch.exp(-x))


def upsample2(x: Tensor) -> Tensor:
    """F.interpolate(scale_factor=2, mode="nearest")"""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def avg_pool2(x: Tensor) -> Tensor:
    """nn.AvgPool2d(2, 2)"""
    b, c, h, w = x.shape
    return x.reshape(b, c, h // 2, 2, w // 2, 2).mean(dim=(3, 5))


def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    half = dim // 2
    freq = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freq[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def residual_block(x: Tensor, t: Optional[Tensor], sd: StateDict, prefix: str = "", eps: float = 1.0e-6,
                   resample: Optional[str] = None, scale_shift: bool = False) -> Tensor:
    """ResidualBlockWithTimeEmbedding._forward (dropout 0); `scale_shift`: residual.py:236-239"""
    inp = x
    net = silu(group_norm(x, sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"], 32, eps))
    if resample == "up":
        inp, net = upsample2(inp), upsample2(net)
    elif resample == "down":
        inp, net = avg_pool2(inp), avg_pool2(net)
    net = CO.conv2d(net, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], 1, 1)
    if prefix + "shortcut.weight" in sd:
        w = sd[prefix + "shortcut.weight"]
        inp = CO.conv2d(inp, w, sd[prefix + "shortcut.bias"], 1, w.shape[-1] // 2)
    if t is not None:
        tt = silu(t) @ sd[prefix + "time_embedding.weight"].t() + sd[prefix + "time_embedding.bias"]
        if scale_shift:
            scale, shift = torch.chunk(tt[:, :, None, None], 2, dim=1)
            net = group_norm(net, sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"], 32, eps) * (1.0 + scale) + shift
            net = CO.conv2d(silu(net), sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], 1, 1)
            return inp + net
        net = net + tt[:, :, None, None]
    net = silu(group_norm(net, sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"], 32, eps))
    net = CO.conv2d(net, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], 1, 1)
    return inp + net


# ---- SpatialTransformer (modules/core/attentions.py:498-569, mixed_stacks/api.py:766-893) ------------------------
import vit_oracle as O  # noqa: E402


def cross_attention(x: Tensor, context: Optional[Tensor], sd: StateDict, prefix: str, num_heads: int) -> Tensor:
    """CrossAttention.forward: to_q / to_k / to_v without bias, heads split as view(B, T, H, dh), softmax(q k^T /
    sqrt(dh)) v, merge, out_linear.0 with bias"""
    ctx = x if context is None else context
    q = x @ sd[prefix + "to_q.weight"].t()
    k = ctx @ sd[prefix + "to_k.weight"].t()
    v = ctx @ sd[prefix + "to_v.weight"].t()
    b, tq, d = q.shape
    dh = d // num_heads
    hd = lambda z: z.reshape(b, z.shape[1], num_heads, dh).permute(0, 2, 1, 3)  # noqa: E731
    o = O.sdp_attention(hd(q), hd(k), hd(v)).permute(0, 2, 1, 3).reshape(b, tq, d)
    return o @ sd[prefix + "out_linear.0.weight"].t() + sd[prefix + "out_linear.0.bias"]


def geglu_feed_forward(x: Tensor, sd: StateDict, prefix: str) -> Tensor:
    """FeedForward(activation="geglu") (channel_mixers.py:25-36; activations.py:150-158)"""
    vg = x @ sd[prefix + "net.0.net.weight"].t() + sd[prefix + "net.0.net.bias"]
    value, gate = vg.chunk(2, dim=-1)
    h = value * O.gelu_erf(gate)
    return h @ sd[prefix + "net.2.linear.weight"].t() + sd[prefix + "net.2.linear.bias"]


def spatial_transformer_block(x: Tensor, context: Optional[Tensor], sd: StateDict, prefix: str, num_heads: int) -> Tensor:
    ln = lambda t, i: O.layer_norm(t, sd[f"{prefix}norm{i}.weight"], sd[f"{prefix}norm{i}.bias"], 1.0e-5)  # noqa: E731
    x = cross_attention(ln(x, 1), None, sd, prefix + "attn1.", num_heads) + x
    x = cross_attention(ln(x, 2), context, sd, prefix + "attn2.", num_heads) + x
    return geglu_feed_forward(ln(x, 3), sd, prefix + "ff.") + x


def spatial_transformer(x: Tensor, context: Optional[Tensor], sd: StateDict, num_heads: int, num_layers: int = 1,
                        prefix: str = "") -> Tensor:
    """SpatialTransformer.forward with use_linear=False: GroupNorm(32, 1e-6) -> 1x1 conv -> tokens -> blocks ->
    1x1 conv -> + input"""
    b, c, h, w = x.shape
    net = group_norm(x, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 32, 1.0e-6)
    wl = sd[prefix + "to_latent.weight"]
    net = net.permute(0, 2, 3, 1).reshape(b, h * w, c) @ wl.reshape(wl.shape[0], -1).t() + sd[prefix + "to_latent.bias"]
    for i in range(num_layers):
        net = spatial_transformer_block(net, context, sd, f"{prefix}blocks.{i}.", num_heads)
    wo = sd[prefix + "from_latent.weight"]
    net = net @ wo.reshape(wo.shape[0], -1).t() + sd[prefix + "from_latent.bias"]
    return x + net.permute(0, 2, 1).reshape(b, c, h, w)


def multi_head_spatial_attention(x: Tensor, sd: StateDict, num_heads: int, split_qkv_before_heads: bool = False,
                                 prefix: str = "") -> Tensor:
    """MultiHeadSpatialAttention._forward (attentions.py:413-460): GroupNorm(32, eps 1e-5) -> conv1d 1x1 to 3C -> heads
    (q | k | v chunks of the 3C channels first, or per head [q | k | v]) -> softmax((q s)^T (k s)) v with s = hd^-1/4 ->
    conv1d 1x1 -> + input"""
    b, c, h, w = x.shape
    area, hd = h * w, c // num_heads
    net = group_norm(x, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 32, 1.0e-5).reshape(b, c, area)
    qkv = torch.einsum("oc,bct->bot", sd[prefix + "to_qkv.weight"][:, :, 0], net) + sd[prefix + "to_qkv.bias"][None, :, None]
    if split_qkv_before_heads:
        q, k, v = (z.reshape(b * num_heads, hd, area) for z in qkv.chunk(3, dim=1))
    else:
        q, k, v = qkv.reshape(b * num_heads, hd * 3, area).split(hd, dim=1)
    scale = 1.0 / math.sqrt(math.sqrt(hd))
    prob = torch.softmax(torch.einsum("bct,bcs->bts", q * scale, k * scale), dim=-1)
    out = torch.einsum("bts,bcs->bct", prob, v).reshape(b, c, area)
    out = torch.einsum("oc,bct->bot", sd[prefix + "to_out.weight"][:, :, 0], out) + sd[prefix + "to_out.bias"][None, :, None]
    return (x.reshape(b, c, area) + out).reshape(b, c, h, w)


def ddpm_objective(pred: Tensor, x: Tensor, noise: Tensor, t: Tensor, tables: Dict[str, Tensor], *,
                   parameterization: str = "eps", loss_type: str = "l2", log_var: Optional[Tensor] = None,
                   l_simple_weight: float = 1.0, original_elbo_weight: float = 0.0) -> Tensor:
    """DDPMStep.loss_fn (models/cv/diffusion.py:44-94): the scalar the trainer differentiates"""
    shape = [-1] + [1] * (x.dim() - 1)
    if parameterization == "eps":
        target = noise
    elif parameterization == "x0":
        target = x
    else:
        target = (tables["sqrt_alphas_cumprod"][t].view(shape) * noise
                  - tables["sqrt_one_minus_alphas_cumprod"][t].view(shape) * x)
    d = pred - target
    loss = (d.abs() if loss_type == "l1" else d * d).mean(dim=(1, 2, 3))
    lv = torch.zeros_like(loss) if log_var is None else log_var[t]
    total = l_simple_weight * (loss / torch.exp(lv) + lv).mean()
    if original_elbo_weight > 0:
        total = total + original_elbo_weight * (tables["lvlb_weights"][t] * loss).mean()
    return total


# ---- UNetDiffuser (modules/multimodal/diffusion/unet.py:76-322), use_spatial_transformer=True ----------------------


def unet_layout(cfg: dict):
    """The block structure the constructor builds (unet.py:206-262): lists of layer kinds per TimestepAttnSequential"""
    mults = tuple(cfg["channel_multipliers"])
    nres = cfg["num_res_blocks"]
    rates = tuple(cfg["attention_downsample_rates"])
    inputs = [["conv"]]
    rate = 1
    for i, _ in enumerate(mults):
        for _ in range(nres):
            inputs.append(["res"] + (["attn"] if rate in rates else []))
        if i != len(mults) - 1:
            inputs.append(["down"])
            rate *= 2
    outputs = []
    for i, _ in list(enumerate(mults))[::-1]:
        for idx in range(nres + 1):
            blk = ["res"] + (["attn"] if rate in rates else [])
            if i != 0 and idx == nres:
                blk.append("up")
                rate //= 2
            outputs.append(blk)
    return inputs, ["res", "attn", "res"], outputs


def unet_diffuser(x: Tensor, timesteps: Tensor, context: Optional[Tensor], sd: StateDict, cfg: dict) -> Tensor:
    """UNetDiffuser.forward (unet.py:268-322) without labels / control"""
    start = cfg["start_channels"]
    heads = cfg["num_heads"]
    nlayers = cfg.get("num_transformer_layers", 1)
    t = timestep_embedding(timesteps, start)
    t = t @ sd["time_embedding.0.weight"].t() + sd["time_embedding.0.bias"]
    t = silu(t) @ sd["time_embedding.2.weight"].t() + sd["time_embedding.2.bias"]

    def run(net: Tensor, kinds, prefix: str) -> Tensor:
        for li, kind in enumerate(kinds):
            p = f"{prefix}{li}."
            if kind == "conv":
                net = CO.conv2d(net, sd[p + "weight"], sd[p + "bias"], 1, 1)
            elif kind == "res":
                net = residual_block(net, t, sd, p, eps=1.0e-5)
            elif kind == "attn":
                net = spatial_transformer(net, context, sd, heads, nlayers, p)
            elif kind == "down":
                net = CO.conv2d(net, sd[p + "net.weight"], sd[p + "net.bias"], 2, 1)
            elif kind == "up":
                net = CO.conv2d(upsample2(net), sd[p + "conv.weight"], sd[p + "conv.bias"], 1, 1)
        return net

    inputs, mid, outputs = unet_layout(cfg)
    nets = []
    net = x
    for bi, kinds in enumerate(inputs):
        net = run(net, kinds, f"input_blocks.{bi}.")
        nets.append(net)
    net = run(net, mid, "residual.")
    for bi, kinds in enumerate(outputs):
        net = run(torch.cat([net, nets.pop()], dim=1), kinds, f"output_blocks.{bi}.")
    net = silu(group_norm(net, sd["head.0.weight"], sd["head.0.bias"], 32, 1.0e-5))
    return CO.conv2d(net, sd["head.2.weight"], sd["head.2.bias"], 1, 1)
