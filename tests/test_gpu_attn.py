"""K3/K4 fused attention vs the CPU oracle (fp32 softmax(q k^T / sqrt(dh)) v on the same
bf16-rounded q, k, v).  Tolerance: the kernel rounds P (and dS) to bf16 before the second MFMA
and the output to bf16 -> rel-L2 <= 1e-2 forward, 2e-2 backward."""
import math

import pytest
import torch

import vit_oracle as O
from helpers import assert_close

pytestmark = pytest.mark.gpu

from cflearn_amd import ops  # noqa: E402

DEV = "cuda"


def _qkv(b, t, h, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(b, t, 3 * h * 64, generator=g) * scale).to(torch.bfloat16)


def _heads(x, h):
    b, t, _ = x.shape
    return x.float().reshape(b, t, h, 64).permute(0, 2, 1, 3)


def _oracle(qkv, h, keep=None, d_o=None):
    d = qkv.shape[-1] // 3
    leaf = qkv.float().requires_grad_(True)
    q, k, v = leaf[..., :d], leaf[..., d:2 * d], leaf[..., 2 * d:]
    b, t, _ = q.shape
    hd = lambda z: z.reshape(b, t, h, 64).permute(0, 2, 1, 3)  # noqa: E731
    o = O.sdp_attention(hd(q), hd(k), hd(v), keep)
    o = o.permute(0, 2, 1, 3).reshape(b, t, d)
    if d_o is not None:
        o.backward(d_o.float())
        return o.detach(), leaf.grad
    return o.detach(), None


CASES = [(2, 197, 12), (3, 17, 2), (1, 256, 1), (2, 64, 3), (2, 1, 2), (1, 33, 4), (2, 130, 2)]


@pytest.mark.parametrize("b,t,h", CASES)
def test_packed_self_attention_fwd_bwd(b, t, h):
    qkv = _qkv(b, t, h, 100 + t, scale=1.5)
    d = h * 64
    d_o = torch.randn(b, t, d, generator=torch.Generator().manual_seed(t)).to(torch.bfloat16)
    want_o, want_g = _oracle(qkv, h, None, d_o)
    dev = qkv.to(DEV)
    o, lse = ops.attn_fwd(dev[..., :d], dev[..., d:2 * d], dev[..., 2 * d:], h)
    assert_close(o, want_o, 1e-2, f"attn fwd {b}x{t}x{h}")
    # lse = natural-log sum-exp of the scaled scores
    s = (_heads(qkv[..., :d], h) @ _heads(qkv[..., d:2 * d], h).transpose(-1, -2)) / 8.0
    assert_close(lse, torch.logsumexp(s, -1), 1e-4, "lse")
    dqkv = torch.zeros_like(dev)
    ops.attn_bwd(dev[..., :d], dev[..., d:2 * d], dev[..., 2 * d:], o, d_o.to(DEV), lse, h,
                 dq=dqkv[..., :d], dk=dqkv[..., d:2 * d], dv=dqkv[..., 2 * d:])
    for nm, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        assert_close(dqkv[..., sl], want_g[..., sl], 2e-2, f"attn {nm} {b}x{t}x{h}", abs_floor=1e-6)


def test_masks_and_causal():
    b, t, h = 3, 50, 2
    d = h * 64
    qkv = _qkv(b, t, h, 7)
    dev = qkv.to(DEV)
    d_o = torch.randn(b, t, d, generator=torch.Generator().manual_seed(8)).to(torch.bfloat16)
    causal_keep = ~torch.triu(torch.ones(t, t, dtype=torch.bool), diagonal=1)
    rnd = torch.rand(b, h, t, t, generator=torch.Generator().manual_seed(9)) < 0.7
    rnd[..., torch.arange(t), torch.arange(t)] = True
    for tag, keep, kw in (("causal flag", causal_keep, dict(causal=True)),
                          ("causal mask", causal_keep, dict(mask=causal_keep.to(DEV))),
                          ("random mask", rnd, dict(mask=rnd.to(DEV)))):
        want_o, want_g = _oracle(qkv, h, keep, d_o)
        o, lse = ops.attn_fwd(dev[..., :d], dev[..., d:2 * d], dev[..., 2 * d:], h, **kw)
        assert_close(o, want_o, 1e-2, f"fwd {tag}")
        dqkv = torch.zeros_like(dev)
        ops.attn_bwd(dev[..., :d], dev[..., d:2 * d], dev[..., 2 * d:], o, d_o.to(DEV), lse, h,
                     dq=dqkv[..., :d], dk=dqkv[..., d:2 * d], dv=dqkv[..., 2 * d:], **kw)
        assert_close(dqkv, want_g, 2e-2, f"bwd {tag}")


def test_golden_sdp(golden):
    """the reference's sdp_attn output frozen in tests/golden/sdp.pt (fp32 inputs, [B,H,T,dh])."""
    g = golden("sdp.pt")
    b, h, t, dh = g["q"].shape
    to_bth = lambda z: z.permute(0, 2, 1, 3).reshape(b, t, h * dh).to(torch.bfloat16).to(DEV)  # noqa: E731
    q, k, v = to_bth(g["q"]), to_bth(g["k"]), to_bth(g["v"])
    o, _ = ops.attn_fwd(q, k, v, h)
    want = g["y_nomask"].permute(0, 2, 1, 3).reshape(b, t, h * dh)
    assert_close(o, want, 1.5e-2, "golden nomask")  # includes the bf16 rounding of q, k, v themselves
    o, _ = ops.attn_fwd(q, k, v, h, mask=g["keep"].to(DEV))
    assert_close(o, g["y_causal"].permute(0, 2, 1, 3).reshape(b, t, h * dh), 1.5e-2, "golden causal")


def test_peaky_softmax_and_cross_lengths():
    """large logits (one key dominating) and Tq != Tk through separate q / kv tensors."""
    b, h = 2, 2
    d = h * 64
    g = torch.Generator().manual_seed(11)
    q = (torch.randn(b, 37, d, generator=g) * 4).to(torch.bfloat16)
    kv = (torch.randn(b, 101, 2, d, generator=g) * 4).to(torch.bfloat16)
    o, lse = ops.attn_fwd(q.to(DEV), kv.to(DEV)[:, :, 0], kv.to(DEV)[:, :, 1], h)
    want = O.sdp_attention(_heads(q, h), _heads(kv[:, :, 0], h), _heads(kv[:, :, 1], h))
    assert_close(o, want.permute(0, 2, 1, 3).reshape(b, 37, d), 1e-2, "cross attention")
    assert torch.isfinite(lse).all()


def test_rows_sum_property_full_size():
    """size-independent property at the bench shape (B=64, T=197, H=12): with v = ones the output
    must be exactly 1 wherever it is defined (softmax rows sum to one), up to bf16 rounding."""
    b, t, h = 64, 197, 12
    d = h * 64
    g = torch.Generator(device=DEV).manual_seed(0)
    qkv = torch.randn(b, t, 3 * d, generator=g, device=DEV).to(torch.bfloat16)
    qkv[..., 2 * d:] = 1.0
    o, _ = ops.attn_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], h)
    assert (o.float() - 1.0).abs().max().item() < 1.0 / 64


LONG_CASES = [(2, 257, 2), (1, 577, 3), (1, 1024, 2), (2, 300, 1)]


@pytest.mark.parametrize("b,t,h", LONG_CASES)
def test_long_sequences_stream_kernels(b, t, h):
    """T > 256 (the chunked / online-softmax kernels): ViT at 384^2 has T = 577, a 32x32 latent 1024 tokens"""
    qkv = _qkv(b, t, h, 300 + t, scale=1.5)
    d = h * 64
    d_o = torch.randn(b, t, d, generator=torch.Generator().manual_seed(t)).to(torch.bfloat16)
    want_o, want_g = _oracle(qkv, h, None, d_o)
    dev = qkv.to(DEV)
    o, lse = ops.attn_fwd(dev[..., :d], dev[..., d:2 * d], dev[..., 2 * d:], h)
    assert_close(o, want_o, 1e-2, f"stream fwd {b}x{t}x{h}")
    s = (_heads(qkv[..., :d], h) @ _heads(qkv[..., d:2 * d], h).transpose(-1, -2)) / 8.0
    assert_close(lse, torch.logsumexp(s, -1), 1e-4, "stream lse")
    for parts in ((3,), (1, 2)):  # both passes in one call, and separately (the dK/dV pass then recomputes delta)
        dqkv = torch.zeros_like(dev)
        for part in parts:
            ops.attn_bwd(dev[..., :d], dev[..., d:2 * d], dev[..., 2 * d:], o, d_o.to(DEV), lse, h,
                         dq=dqkv[..., :d], dk=dqkv[..., d:2 * d], dv=dqkv[..., 2 * d:], parts=part)
        for nm, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
            assert_close(dqkv[..., sl], want_g[..., sl], 2e-2, f"stream {nm} {b}x{t}x{h} parts {parts}", abs_floor=1e-6)


def test_long_sequences_masks_causal_and_cross_lengths():
    b, t, h = 2, 333, 2
    d = h * 64
    qkv = _qkv(b, t, h, 17)
    dev = qkv.to(DEV)
    d_o = torch.randn(b, t, d, generator=torch.Generator().manual_seed(18)).to(torch.bfloat16)
    causal_keep = ~torch.triu(torch.ones(t, t, dtype=torch.bool), diagonal=1)
    rnd = torch.rand(b, h, t, t, generator=torch.Generator().manual_seed(19)) < 0.7
    rnd[..., torch.arange(t), torch.arange(t)] = True
    for tag, keep, kw in (("causal flag", causal_keep, dict(causal=True)),
                          ("random mask", rnd, dict(mask=rnd.to(DEV)))):
        want_o, want_g = _oracle(qkv, h, keep, d_o)
        o, lse = ops.attn_fwd(dev[..., :d], dev[..., d:2 * d], dev[..., 2 * d:], h, **kw)
        assert_close(o, want_o, 1e-2, f"stream fwd {tag}")
        dqkv = torch.zeros_like(dev)
        ops.attn_bwd(dev[..., :d], dev[..., d:2 * d], dev[..., 2 * d:], o, d_o.to(DEV), lse, h,
                     dq=dqkv[..., :d], dk=dqkv[..., d:2 * d], dv=dqkv[..., 2 * d:], **kw)
        assert_close(dqkv, want_g, 2e-2, f"stream bwd {tag}")
    # Tq short, Tk long (cross attention over a long context) and the reverse
    g = torch.Generator().manual_seed(21)
    for tq, tk in ((37, 700), (513, 40)):
        q = (torch.randn(b, tq, d, generator=g) * 2).to(torch.bfloat16)
        kv = (torch.randn(b, tk, 2, d, generator=g) * 2).to(torch.bfloat16)
        o, lse = ops.attn_fwd(q.to(DEV), kv.to(DEV)[:, :, 0], kv.to(DEV)[:, :, 1], h)
        want = O.sdp_attention(_heads(q, h), _heads(kv[:, :, 0], h), _heads(kv[:, :, 1], h))
        assert_close(o, want.permute(0, 2, 1, 3).reshape(b, tq, d), 1e-2, f"stream cross {tq}x{tk}")


def _oracle_dh(q, k, v, h, dh, d_o, keep=None):
    ql, kl, vl = (t.float().requires_grad_(True) for t in (q, k, v))
    b, tq, _ = q.shape
    tk = k.shape[1]
    hd = lambda z, t: z.reshape(b, t, h, dh).permute(0, 2, 1, 3)  # noqa: E731
    o = O.sdp_attention(hd(ql, tq), hd(kl, tk), hd(vl, tk), keep)
    o = o.permute(0, 2, 1, 3).reshape(b, tq, h * dh)
    o.backward(d_o.float())
    return o.detach(), ql.grad, kl.grad, vl.grad


@pytest.mark.parametrize("dh,tq,tk,h", [(40, 64, 64, 8), (40, 300, 77, 2), (80, 256, 256, 3), (160, 64, 64, 2),
                                         (160, 130, 300, 1), (64, 100, 77, 2), (8, 20, 33, 2), (96, 17, 17, 2)])
def test_general_head_dims(dh, tq, tk, h):
    """head_dim != 64 (the UNet's 40 / 80 / 160-channel heads) and Tq != Tk (cross attention over a context)"""
    g = torch.Generator().manual_seed(dh * 1000 + tq)
    d = h * dh
    q = (torch.randn(2, tq, d, generator=g) * 1.2).to(torch.bfloat16)
    k = (torch.randn(2, tk, d, generator=g) * 1.2).to(torch.bfloat16)
    v = (torch.randn(2, tk, d, generator=g) * 1.2).to(torch.bfloat16)
    d_o = torch.randn(2, tq, d, generator=g).to(torch.bfloat16)
    want_o, gq, gk, gv = _oracle_dh(q, k, v, h, dh, d_o)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o, lse = ops.attn_fwd(qd, kd, vd, h, head_dim=dh)
    assert_close(o, want_o, 1e-2, f"fwd dh={dh}")
    dq, dk, dv = torch.zeros_like(qd), torch.zeros_like(kd), torch.zeros_like(vd)
    ops.attn_bwd(qd, kd, vd, o, d_o.to(DEV), lse, h, dq=dq, dk=dk, dv=dv, head_dim=dh)
    assert_close(dq, gq, 2e-2, f"dq dh={dh}", abs_floor=1e-6)
    assert_close(dk, gk, 2e-2, f"dk dh={dh}", abs_floor=1e-6)
    assert_close(dv, gv, 2e-2, f"dv dh={dh}", abs_floor=1e-6)
