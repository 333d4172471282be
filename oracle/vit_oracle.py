"""CPU oracle for the data-parallel training hot path (ViT building blocks).

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this file; the product path
(`carefree-learn_amd/`) never does and fails loudly when its HIP library is missing.

What it is: a plain fp32 (or fp64) restatement, in elementary tensor arithmetic on the CPU, of
the algorithm the reference's `cflearn.modules` blocks execute on the ViT path.  Each function
cites the reference file:line (relative to /root/reference) it follows.  Parameters are taken
from a flat `state_dict` that uses the REFERENCE'S key names, so the same dict can be loaded
into the reference modules, into this oracle and into the HIP modules.

Pinning: the reference has no stored golden files; its own known-answer tests
compare against PyTorch CPU ops with injected identical weights.  This oracle is pinned the same
way, against the reference modules THEMSELVES imported from /root/reference through
`oracle/refharness` (build container only): `tests/test_oracle_vs_reference.py` runs that check
live when the reference tree is present, and `oracle/gen_golden.py` freezes reference outputs
into `tests/golden/*.pt` so the pin also holds on the GPU box (where /root/reference is absent).
"""
import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

StateDict = Dict[str, Tensor]

# ---------------------------------------------------------------------------------------------
# element-wise / normalisation
# ---------------------------------------------------------------------------------------------


def gelu_erf(x: Tensor) -> Tensor:
    """Exact-erf GELU: `build_activation("GELU")` -> `nn.GELU()` (activations.py:46-48)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def quick_gelu(x: Tensor) -> Tensor:
    """`x * sigmoid(1.702 x)` (activations.py:150-153)."""
    return x * torch.sigmoid(1.702 * x)


def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1.0e-6) -> Tensor:
    """`nn.LayerNorm(D, eps)`: biased variance, eps inside the sqrt.

    eps defaults to 1e-6 because `NormFactory("layer")` injects it (norms.py:118-119).
    """
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * weight + bias


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """`F.linear(net, W, b)` = x @ W^T + b  (customs.py:89)."""
    y = x @ weight.transpose(-1, -2)
    if bias is not None:
        y = y + bias
    return y


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------


def sdp_attention(q: Tensor, k: Tensor, v: Tensor, keep_mask: Optional[Tensor] = None) -> Tensor:
    """softmax(q k^T / sqrt(dh) [masked]) v   (toolkit.py:959-974).

    q, k, v: [B, H, T, dh].  `keep_mask` (bool, broadcastable to [B, H, Tq, Tk]): True = attend,
    i.e. the polarity AFTER the inversion at attentions.py:252-253.  The scale is always
    1/sqrt(dh) on this path; the module's `qk_scale` only affects the slow path.
    """
    dh = q.shape[-1]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if keep_mask is not None:
        s = s.masked_fill(~keep_mask, float("-inf"))
    m = s.max(dim=-1, keepdim=True).values
    p = torch.exp(s - m)
    p = p / p.sum(dim=-1, keepdim=True)
    return p @ v


def expand_module_mask(mask: Tensor, num_heads: int) -> Tensor:
    """The mask layout quirk of `Attention.forward` (attentions.py:246-249).

    Module input `mask` is [B, Tq, Tk] with True = slot zeroed.  The reference does
    `mask.repeat(H, 1, 1).view(-1, H, Tq, Tk)`, so the effective mask of (b, h) is
    `mask[(b * H + h) % B]`.  Returns the bool KEEP mask [B, H, Tq, Tk].
    """
    b, tq, tk = mask.shape
    idx = (torch.arange(b * num_heads) % b).view(b, num_heads)
    return ~mask[idx]


def self_attention(
    x: Tensor,
    sd: StateDict,
    prefix: str,
    num_heads: int,
    mask: Optional[Tensor] = None,
) -> Tensor:
    """`Attention(is_self_attention=True)` forward, fast path (attentions.py:198-279).

    Packed projection `in_w` [3D, Din]: rows [0:D]=q, [D:2D]=k, [2D:3D]=v (chunk(3), :214-216);
    head h owns channels [h*dh:(h+1)*dh] (`view(B,T,H,dh).permute(0,2,1,3)`, :180-185);
    heads are merged back as [B, T, H*dh] (:270-275) and sent through `out_linear` (:277).
    """
    b, t, _ = x.shape
    in_w = sd[prefix + "in_w"]
    qkv_bias = sd.get(prefix + "qkv_bias")
    d = in_w.shape[0] // 3
    dh = d // num_heads
    qkv = linear(x, in_w, qkv_bias)
    q, k, v = qkv[..., :d], qkv[..., d : 2 * d], qkv[..., 2 * d :]

    def heads(z: Tensor) -> Tensor:
        return z.reshape(b, t, num_heads, dh).permute(0, 2, 1, 3)

    keep = None if mask is None else expand_module_mask(mask, num_heads)
    o = sdp_attention(heads(q), heads(k), heads(v), keep)
    o = o.permute(0, 2, 1, 3).reshape(b, t, d)
    return linear(o, sd[prefix + "out_linear.linear.weight"], sd.get(prefix + "out_linear.linear.bias"))


# ---------------------------------------------------------------------------------------------
# feed-forward / mixing block
# ---------------------------------------------------------------------------------------------


def feed_forward(x: Tensor, sd: StateDict, prefix: str, activation: str = "GELU") -> Tensor:
    """`FeedForward`: Linear -> act -> Dropout(0) -> Linear -> Dropout(0) (channel_mixers.py:15-43).

    state keys `net.0.linear.*` and `net.3.linear.*`.
    """
    h = linear(x, sd[prefix + "net.0.linear.weight"], sd.get(prefix + "net.0.linear.bias"))
    if activation == "GELU":
        h = gelu_erf(h)
    elif activation == "quick_gelu":
        h = quick_gelu(h)
    else:
        raise NotImplementedError(activation)
    return linear(h, sd[prefix + "net.3.linear.weight"], sd.get(prefix + "net.3.linear.bias"))


def mixing_block(
    x: Tensor,
    sd: StateDict,
    prefix: str,
    num_heads: int,
    eps: float = 1.0e-6,
    mask: Optional[Tensor] = None,
) -> Tensor:
    """Pre-norm `MixingBlock` with attention token mixer + FF channel mixer, no dropout / drop-path
    (mixed_stacks/api.py:130-158):  x1 = x + attn(LN(x));  x2 = x1 + ff(LN(x1)).
    """
    n1 = layer_norm(x, sd[prefix + "token_norm.weight"], sd[prefix + "token_norm.bias"], eps)
    x = x + self_attention(n1, sd, prefix + "token_mixing.net.", num_heads, mask)
    n2 = layer_norm(x, sd[prefix + "channel_norm.weight"], sd[prefix + "channel_norm.bias"], eps)
    return x + feed_forward(n2, sd, prefix + "channel_mixing.")


def mixing_block_post_norm(x: Tensor, sd: StateDict, prefix: str, num_heads: int, eps: float = 1.0e-6) -> Tensor:
    """Post-norm `MixingBlock` (mixed_stacks/api.py:160-185): x1 = LN(x + attn(x)); x2 = LN(x1 + ff(x1))."""
    x = x + self_attention(x, sd, prefix + "token_mixing.net.", num_heads)
    x = layer_norm(x, sd[prefix + "token_norm.weight"], sd[prefix + "token_norm.bias"], eps)
    x = x + feed_forward(x, sd, prefix + "channel_mixing.")
    return layer_norm(x, sd[prefix + "channel_norm.weight"], sd[prefix + "channel_norm.bias"], eps)


def interpolate_pos_encoding(pos: Tensor, num_head_tokens: int, num_current: int, h: int, w: int) -> Tensor:
    """`PositionalEncoding.interpolate_pos_encoding` (mixed_stacks/api.py:231-267): bicubic resample of the learned
    [sqrt(T) x sqrt(T)] grid (align_corners=False, recompute_scale_factor=True, toolkit.py:2841-2861) to the patch
    grid of an `h x w` image; head-token rows pass through."""
    import math

    head, grid = pos[:, :num_head_tokens], pos[:, num_head_tokens:]
    num_history, dim = grid.shape[1], pos.shape[-1]
    sqrt = math.sqrt(num_history)
    wh_ratio = w / h
    pw = math.sqrt(num_current * wh_ratio) + 0.1
    ph = math.sqrt(num_current / wh_ratio) + 0.1
    grid = torch.nn.functional.interpolate(grid.reshape(1, int(sqrt), int(sqrt), dim).permute(0, 3, 1, 2), mode="bicubic",
                                           scale_factor=(pw / sqrt, ph / sqrt), recompute_scale_factor=True,
                                           align_corners=False)
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat([head, grid], dim=1)


# ---------------------------------------------------------------------------------------------
# ViT encoder + classifier
# ---------------------------------------------------------------------------------------------


def patch_embed(img: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """`VanillaPatchEmbed`: Conv2d(k = stride = patch, pad 0) then flatten(2).transpose(1, 2)
    (high_level.py:172-188, 144-149).  Restated as im2row + matmul: the conv at stride == kernel
    is a dot product of each non-overlapping patch (c, ph, pw order) with W[out, c, ph, pw]."""
    b, c, hh, ww = img.shape
    p = weight.shape[-1]
    gh, gw = hh // p, ww // p
    rows = img.reshape(b, c, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, c * p * p)
    return linear(rows, weight.reshape(weight.shape[0], -1), bias)


def vit_encoder(
    img: Tensor,
    sd: StateDict,
    num_heads: int,
    num_layers: int,
    prefix: str = "",
    eps: float = 1.0e-6,
) -> Tensor:
    """`ViTEncoder.forward` (cv/encoder/transformer.py:88-100) with the defaults of the ViT path:
    head token, learned positional encoding at native resolution, pre-norm blocks, head =
    `PreNorm(LN) -> x[:, 0]` (mixed_stacks/api.py:363-402,419-458).  Returns [B, D]."""
    x = patch_embed(
        img, sd[prefix + "to_patches.projection.weight"], sd.get(prefix + "to_patches.projection.bias")
    )
    b = x.shape[0]
    head_token = sd[prefix + "encoder.head_token"]
    x = torch.cat([head_token.expand(b, -1, -1), x], dim=1)
    x = x + sd[prefix + "encoder.pos_encoding.pos_encoding"]
    for i in range(num_layers):
        x = mixing_block(x, sd, f"{prefix}encoder.mixing_blocks.{i}.", num_heads, eps)
    x = layer_norm(
        x, sd[prefix + "encoder.head.norms.0.weight"], sd[prefix + "encoder.head.norms.0.bias"], eps
    )
    x = x[:, 0]
    proj = sd.get(prefix + "output_projection")
    if proj is not None:
        x = x @ proj
    return x


def vit_classifier(img: Tensor, sd: StateDict, num_heads: int, num_layers: int) -> Tensor:
    """`head(ViTEncoder(x))` composed by hand, as SURVEY F6 prescribes (`cv_clf(encoder="vit")`
    crashes in the reference).  Keys: `encoder.*` for the ViT encoder, `head.linear.*` for the
    `Linear(latent, num_classes)` head (cv/classifier/vanilla.py:43,61)."""
    z = vit_encoder(img, sd, num_heads, num_layers, prefix="encoder.")
    return linear(z, sd["head.linear.weight"], sd.get("head.linear.bias"))


# ---------------------------------------------------------------------------------------------
# losses (integer label gather: bit-exact class for the index part)
# ---------------------------------------------------------------------------------------------


def cross_entropy(logits: Tensor, labels: Tensor) -> Tensor:
    """mean over the batch of -log softmax(logits)[label]  (losses/basic.py:126-141)."""
    labels = labels.view(-1)
    m = logits.max(dim=1, keepdim=True).values
    lse = m.squeeze(1) + torch.log(torch.exp(logits - m).sum(dim=1))
    picked = logits.gather(1, labels.view(-1, 1)).squeeze(1)
    return (lse - picked).mean()


def focal_loss(logits: Tensor, labels: Tensor, gamma: float = 2.0, eps: float = 1.0e-6) -> Tensor:
    """p = softmax(logits) + 1e-6; loss = -log(p_y) * (1 - p_y)^gamma, mean  (losses/basic.py:170-206)."""
    p = torch.softmax(logits, dim=1) + eps
    py = p.gather(1, labels.view(-1, 1)).squeeze(1)
    return (-torch.log(py) * (1.0 - py) ** gamma).mean()


# ---------------------------------------------------------------------------------------------
# one optimisation step (what bench.py's cpu_baseline times and what DDP parity is defined on)
# ---------------------------------------------------------------------------------------------


def loss_and_grads(
    img: Tensor,
    labels: Tensor,
    sd: StateDict,
    num_heads: int,
    num_layers: int,
    loss: str = "cross_entropy",
) -> Tuple[Tensor, Tensor, StateDict]:
    """Forward + loss + backward of the ViT classifier in fp32 on the CPU.

    Returns (loss, logits, grads keyed like `sd`)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits = vit_classifier(img, leaves, num_heads, num_layers)
    fn = cross_entropy if loss == "cross_entropy" else focal_loss
    value = fn(logits, labels)
    value.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return value.detach(), logits.detach(), grads


def adamw_step(
    p: Tensor,
    g: Tensor,
    m: Tensor,
    v: Tensor,
    step: int,
    lr: float,
    beta1: float = 0.9,
    beta2: float = 0.999,
    eps: float = 1.0e-8,
    weight_decay: float = 0.0,
    decoupled: bool = True,
) -> None:
    """In-place Adam / AdamW update as `torch.optim.Adam(W)` defines it (the reference's optimizer
    registry maps "adam"/"adamw" straight to those, optimizers.py:29-33)."""
    if weight_decay != 0.0:
        if decoupled:
            p.mul_(1.0 - lr * weight_decay)
        else:
            g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1**step
    bc2 = 1.0 - beta2**step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
