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
                                         (160, 130, 300, 1), (64, 100, 77, 2), (8, 20, 33, 2), (96, 17, 17, 2),
                                         (72, 200, 333, 2), (88, 150, 260, 1), (56, 90, 200, 2)])
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


# ---------------------------------------------------------------------------------------------
# dropout on the attention probabilities (reference attentions.py:254 -> sdp_attn(..., dropout))
# ---------------------------------------------------------------------------------------------


def _dropout_reference(q, k, v, h, dh, keep_prob_mask, p_eff, user_keep=None, causal=False, d_o=None):
    """fp32 autograd of O = (softmax(S) * M / (1 - p)) V for a GIVEN mask M [B, H, Tq, Tk]"""
    leaves = [t.float().requires_grad_(True) for t in (q, k, v)]
    b, tq, _ = q.shape
    tk = k.shape[1]
    hd = lambda z, t: z.reshape(b, t, h, dh).permute(0, 2, 1, 3)  # noqa: E731
    s = hd(leaves[0], tq) @ hd(leaves[1], tk).transpose(-1, -2) / math.sqrt(dh)
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(tq, tk, dtype=torch.bool), 1), float("-inf"))
    if user_keep is not None:
        s = s.masked_fill(user_keep == 0, float("-inf"))
    prob = torch.softmax(s, -1) * (keep_prob_mask.float() / (1.0 - p_eff))
    o = (prob @ hd(leaves[2], tk)).permute(0, 2, 1, 3).reshape(b, tq, h * dh)
    if d_o is not None:
        o.backward(d_o.float())
        return o.detach(), [t.grad for t in leaves]
    return o.detach(), None


@pytest.mark.parametrize("dh,tq,tk,h,causal", [(64, 197, 197, 3, False), (64, 70, 90, 2, False), (40, 130, 77, 2, False),
                                               (64, 300, 300, 1, True), (128, 50, 260, 2, False)])
def test_attention_probability_dropout_matches_the_formula_for_its_own_mask(dh, tq, tk, h, causal):
    """O = dropout(softmax(S)) V and all three gradients, against fp32 autograd with the mask the kernels derive from
    (seed, offset) — exported by cfhip_attn_dropout_mask — for resident-size and chunked shapes, head widths 40 / 64 /
    128, cross lengths, causal and a user keep-mask.  p is quantised to 1/256: 0.3 -> 77/256."""
    b, p = 2, 0.3
    p_eff = round(p * 256) / 256
    g = torch.Generator().manual_seed(dh + tq)
    q = torch.randn(b, tq, h * dh, generator=g).to(torch.bfloat16)
    k = torch.randn(b, tk, h * dh, generator=g).to(torch.bfloat16)
    v = torch.randn(b, tk, h * dh, generator=g).to(torch.bfloat16)
    d_o = torch.randn(b, tq, h * dh, generator=g).to(torch.bfloat16)
    user_keep = None
    if dh == 40:  # with a user mask too (every row keeps its first key)
        user_keep = (torch.rand(b, h, tq, tk, generator=g) > 0.4).to(torch.uint8)
        user_keep[..., 0] = 1
    seed, offset = 1234567, 4096
    mask = ops.attn_dropout_mask(b, h, tq, tk, p, seed, offset)
    frac = mask.float().mean().item()
    n = mask.numel()
    assert abs(frac - (1 - p_eff)) < 5 * math.sqrt(p_eff * (1 - p_eff) / n), (frac, 1 - p_eff)
    want_o, want_g = _dropout_reference(q, k, v, h, dh, mask.cpu(), p_eff, user_keep, causal, d_o)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    ukd = None if user_keep is None else user_keep.to(DEV)
    kw = dict(mask=ukd, causal=causal, head_dim=dh, dropout_p=p, seed=seed, offset=offset)
    o, lse = ops.attn_fwd(qd, kd, vd, h, **kw)
    assert_close(o, want_o, 1e-2, "dropout attention fwd")
    o2, _ = ops.attn_fwd(qd, kd, vd, h, **kw)
    assert torch.equal(o, o2)  # a pure function of (seed, offset)
    o3, _ = ops.attn_fwd(qd, kd, vd, h, **dict(kw, offset=offset + ops.attn_dropout_blocks(b, h, tq, tk)))
    assert not torch.equal(o, o3)
    dq, dk, dv = torch.zeros_like(qd), torch.zeros_like(kd), torch.zeros_like(vd)
    ops.attn_bwd(qd, kd, vd, o, d_o.to(DEV), lse, h, dq=dq, dk=dk, dv=dv, **kw)
    assert_close(dq, want_g[0], 2e-2, "dropout attention dq")
    assert_close(dk, want_g[1], 2e-2, "dropout attention dk")
    assert_close(dv, want_g[2], 2e-2, "dropout attention dv")
    # the two passes launched separately (two streams in the step) see the same mask
    dq2, dk2, dv2 = torch.zeros_like(qd), torch.zeros_like(kd), torch.zeros_like(vd)
    ops.attn_bwd(qd, kd, vd, o, d_o.to(DEV), lse, h, dq=dq2, dk=dk2, dv=dv2, parts=1, **kw)
    ops.attn_bwd(qd, kd, vd, o, d_o.to(DEV), lse, h, dq=dq2, dk=dk2, dv=dv2, parts=2, **kw)
    assert torch.equal(dq, dq2) and torch.equal(dv, dv2)
    assert_close(dk2, dk, 2e-3, "dk of the stand-alone pass")  # (it recomputes delta = rowsum(dO * O) in another order)


def test_attention_module_dropout_trains_and_is_off_in_eval():
    import cflearn_amd as C

    torch.manual_seed(0)
    m = C.Attention(128, 2, dropout=0.2).to(DEV)  # head_dim 64: packed path
    x = torch.randn(3, 50, 128, device=DEV)
    m.eval()
    y_eval = m(x, x, x).output
    m.dropout = 0.0
    assert torch.equal(m(x, x, x).output, y_eval)
    m.dropout = 0.2
    m.train()
    ops.PhiloxState.manual_seed(5)
    off0 = ops.PhiloxState.offset
    xr = x.clone().requires_grad_(True)
    y1 = m(xr, xr, xr).output
    assert ops.PhiloxState.offset - off0 == ops.attn_dropout_blocks(3, 2, 50, 50)
    y2 = m(x, x, x).output
    assert not torch.equal(y1, y2) and not torch.equal(y1, y_eval)  # fresh mask every call
    y1.float().sum().backward()
    assert torch.isfinite(xr.grad).all() and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    ops.PhiloxState.manual_seed(5)
    assert torch.equal(m(x, x, x).output, y1)  # reseeding reproduces the run
    # E[dropout(P)] = P: the train-mode output averages to the eval-mode one
    acc = torch.zeros_like(y_eval, dtype=torch.float32)
    for _ in range(600):
        acc += m(x, x, x).output.float()
    assert_close(acc / 600, y_eval.float(), 6e-2, "mean over dropout masks")


@pytest.mark.parametrize("dh,t,h", [(80, 1024, 3), (72, 300, 2), (96, 1000, 2), (88, 777, 1), (80, 129, 2)])
def test_two_tile_forms_of_head_dim_72_to_96(dh, t, h):
    """`attn_fwd2_kernel<true, 5 / 6>` and `attn_bwd_dq2_kernel<true, 5 / 6>` (round 6: option "attn_two_tiles" bit 256, on by default — the
    UNet's 80-channel heads) against the one-tile general kernels they replace (bit off) and against fp32 autograd: ragged lengths, a
    length just beyond one chunk, the widest head_dim of the form; delta as the dQ pass writes it."""
    b = 2
    d = h * dh
    g = torch.Generator(device=DEV).manual_seed(dh * 7 + t)
    rnd = lambda *s: (torch.randn(*s, generator=g, device=DEV) * 0.8).to(torch.bfloat16)  # noqa: E731
    q, k, v, d_o = rnd(b, t, d), rnd(b, t, d), rnd(b, t, d), rnd(b, t, d)
    leaves = [z.float().requires_grad_(True) for z in (q, k, v)]
    hd = lambda z: z.reshape(b, t, h, dh).permute(0, 2, 1, 3)  # noqa: E731
    s = (hd(leaves[0]) @ hd(leaves[1]).transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    want_lse = torch.logsumexp(s.detach(), -1)
    want = (torch.softmax(s, -1) @ hd(leaves[2])).permute(0, 2, 1, 3).reshape(b, t, d)
    want.backward(d_o.float())
    outs = []
    for bits in (255, 511):
        ops.set_option("attn_two_tiles", bits)
        try:
            o, lse = ops.attn_fwd(q, k, v, h, head_dim=dh)
            dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
            delta = torch.zeros(b, h, t, device=DEV)
            ops.attn_bwd(q, k, v, o, d_o, lse, h, dq=dq, dk=dk, dv=dv, delta=delta, head_dim=dh)
            outs.append((o, lse, dq, dk, dv, delta))
        finally:
            ops.set_option("attn_two_tiles", 511)
    for tag, (o, lse, dq, dk, dv, delta) in zip(("one tile", "two tiles"), outs):
        assert_close(o, want.detach(), 1e-2, f"{tag}: fwd dh={dh}")
        assert_close(lse, want_lse, 1e-4, f"{tag}: lse dh={dh}")
        for nm, got, leaf in (("dq", dq, leaves[0]), ("dk", dk, leaves[1]), ("dv", dv, leaves[2])):
            assert_close(got, leaf.grad, 2e-2, f"{tag}: {nm} dh={dh}", abs_floor=1e-6)
    assert_close(outs[1][5], outs[0][5], 2e-3, "delta of the two dQ passes", abs_floor=1e-4)
    assert_close(outs[1][0], outs[0][0].float(), 4e-3, "forward of the two forms", abs_floor=1e-6)


@pytest.mark.parametrize("t,dh", [(4096, 40), (4096, 80), (16384, 40), (16384, 80), (4096, 160), (4096, 64), (5000, 48)])
def test_attention_at_the_unet_token_counts(t, dh):
    """The general-length kernels at the sizes the DDPM UNet runs them (BASELINE config 4: `SpatialTransformer` self
    attention over 64^2 / 128^2 pixels with 40- / 80- / 160-channel heads — mixed_stacks/api.py:766-893,
    attentions.py:498-569; VERDICT r2: untested beyond T = 1024): output, log-sum-exp and all three gradients against fp32
    torch autograd of softmax(q k^T / sqrt(dh)) v on the same bf16 operands, computed on the device (the T x T scores of two
    heads fit in HBM: 2 GB at T = 16 384)."""
    b, h = 1, 2
    d = h * dh
    g = torch.Generator(device=DEV).manual_seed(t + dh)
    rnd = lambda *s: torch.randn(*s, generator=g, device=DEV).to(torch.bfloat16)  # noqa: E731
    q, k, v, d_o = rnd(b, t, d), rnd(b, t, d), rnd(b, t, d), rnd(b, t, d)
    leaves = [z.float().requires_grad_(True) for z in (q, k, v)]
    hd = lambda z: z.reshape(b, t, h, dh).permute(0, 2, 1, 3)  # noqa: E731
    s = (hd(leaves[0]) @ hd(leaves[1]).transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    want_lse = torch.logsumexp(s.detach(), -1)
    want = (torch.softmax(s, -1) @ hd(leaves[2])).permute(0, 2, 1, 3).reshape(b, t, d)
    del s
    want.backward(d_o.float())
    o, lse = ops.attn_fwd(q, k, v, h, head_dim=dh)
    assert_close(o, want.detach(), 1e-2, f"fwd T={t} dh={dh}")
    assert_close(lse, want_lse, 1e-4, f"lse T={t} dh={dh}")
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    ops.attn_bwd(q, k, v, o, d_o, lse, h, dq=dq, dk=dk, dv=dv, head_dim=dh)
    for nm, got, leaf in (("dq", dq, leaves[0]), ("dk", dk, leaves[1]), ("dv", dv, leaves[2])):
        assert_close(got, leaf.grad, 2e-2, f"{nm} T={t} dh={dh}", abs_floor=1e-6)


def test_attention_at_the_256px_unet_level_sampled():
    """T = 65 536 tokens, 8 heads of 40 channels, batch 1: the full-resolution self-attention of the DDPM UNet at 256^2
    (BASELINE config 4; mixed_stacks/api.py:766-893 -> attentions.py:498-569 -> toolkit.py:911-974) — the three kernels that
    are 72 % of that step (`attn_fwd2_kernel<true,3,true>`, `attn_bwd_dq2_kernel<true,3>`, `attn_bwd_dkv2_kernel<true,3>`) and
    were compared with fp32 math only up to T = 16 384.  The T x T scores do not fit as one tensor: the fp32
    reference walks the query rows in chunks of 1 024 (output and log-sum-exp of EVERY row), dQ is checked on 256 sampled
    query rows against all 65 536 keys, dK / dV on 256 sampled keys against all 65 536 queries."""
    b, h, t, dh = 1, 8, 65536, 40
    d = h * dh
    scale = 1.0 / math.sqrt(dh)
    g = torch.Generator(device=DEV).manual_seed(65536)
    rnd = lambda *s: torch.randn(*s, generator=g, device=DEV).to(torch.bfloat16)  # noqa: E731
    q, k, v, d_o = rnd(b, t, d), rnd(b, t, d), rnd(b, t, d), rnd(b, t, d)
    o, lse = ops.attn_fwd(q, k, v, h, head_dim=dh)
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    ops.attn_bwd(q, k, v, o, d_o, lse, h, dq=dq, dk=dk, dv=dv, head_dim=dh)

    hd = lambda z: z.float().reshape(t, h, dh).permute(1, 0, 2).contiguous()  # noqa: E731  [h, T, dh]
    qf, kf, vf, dof = hd(q[0]), hd(k[0]), hd(v[0]), hd(d_o[0])
    want_o = torch.empty(h, t, dh, device=DEV)
    want_lse = torch.empty(h, t, device=DEV)
    for r0 in range(0, t, 1024):
        s = (qf[:, r0:r0 + 1024] @ kf.transpose(1, 2)) * scale          # [h, 1024, T] fp32: 2 GB
        want_lse[:, r0:r0 + 1024] = torch.logsumexp(s, -1)
        want_o[:, r0:r0 + 1024] = torch.softmax(s, -1) @ vf
        del s
    assert_close(o[0].float().reshape(t, h, dh).permute(1, 0, 2), want_o, 1e-2, "fwd T=65536 dh=40")
    assert_close(lse[0], want_lse, 1e-4, "lse T=65536 dh=40")
    delta = (dof * want_o).sum(-1)                                        # [h, T]

    rows = torch.randperm(t, generator=g, device=DEV)[:256].sort().values
    s = (qf[:, rows] @ kf.transpose(1, 2)) * scale                        # [h, 256, T]
    p = torch.exp(s - want_lse[:, rows, None])
    ds = p * (dof[:, rows] @ vf.transpose(1, 2) - delta[:, rows, None])
    want_dq = (ds @ kf) * scale                                           # [h, 256, dh]
    got_dq = dq[0].float().reshape(t, h, dh)[rows].permute(1, 0, 2)
    assert_close(got_dq, want_dq, 2e-2, "dq (256 sampled query rows, all keys)", abs_floor=1e-6)
    del s, p, ds

    keys = torch.randperm(t, generator=g, device=DEV)[:256].sort().values
    s = (qf @ kf[:, keys].transpose(1, 2)) * scale                        # [h, T, 256]
    p = torch.exp(s - want_lse[:, :, None])
    want_dv = p.transpose(1, 2) @ dof                                     # [h, 256, dh]
    ds = p * (dof @ vf[:, keys].transpose(1, 2) - delta[:, :, None])
    want_dk = (ds.transpose(1, 2) @ qf) * scale
    got_dk = dk[0].float().reshape(t, h, dh)[keys].permute(1, 0, 2)
    got_dv = dv[0].float().reshape(t, h, dh)[keys].permute(1, 0, 2)
    assert_close(got_dv, want_dv, 2e-2, "dv (256 sampled keys, all queries)", abs_floor=1e-6)
    assert_close(got_dk, want_dk, 2e-2, "dk (256 sampled keys, all queries)", abs_floor=1e-6)


def _weights_reference(q, k, v, h, dh, scale, keep=None, causal=False):
    """the reference's slow path (attentions.py:256-268) in fp32 on the same bf16 operands: raw = q k^T * scale,
    masked_fill(-inf), softmax, weights @ v; returns leaves + (output [B, Tq, H dh], weights [B, H, Tq, Tk])"""
    leaves = [t.float().requires_grad_(True) for t in (q, k, v)]
    b, tq, _ = q.shape
    tk = k.shape[1]
    hd = lambda z, t: z.reshape(b, t, h, dh).permute(0, 2, 1, 3)  # noqa: E731
    raw = (hd(leaves[0], tq) @ hd(leaves[1], tk).transpose(-1, -2)) * scale
    drop = torch.zeros(b, h, tq, tk, dtype=torch.bool)
    if keep is not None:
        drop |= ~keep.bool().expand(b, h, tq, tk)
    if causal:
        drop |= torch.ones(tq, tk, dtype=torch.bool).triu(1)
    w = torch.softmax(raw.masked_fill(drop, float("-inf")), dim=-1)
    o = (w @ hd(leaves[2], tk)).permute(0, 2, 1, 3).reshape(b, tq, h * dh)
    return leaves, o, w


@pytest.mark.parametrize("dh,tq,tk,h,causal,masked,scale", [(64, 197, 197, 3, False, False, None), (40, 50, 77, 2, False, False, 0.2),
                                                              (64, 33, 33, 4, True, False, None), (80, 70, 90, 2, False, True, None),
                                                              (64, 300, 300, 2, False, True, 0.1), (8, 5, 3, 1, False, False, None)])
def test_attention_with_returned_weights(dh, tq, tk, h, causal, masked, scale):
    """`Attention.forward(require_weights=True)` at the kernel level: output AND weights against the reference's slow-path
    formula in fp32, and the gradients of a loss that uses BOTH (the weights' share goes through cfhip_attn_probs_bwd)."""
    from cflearn_amd import functional as HF

    b = 2
    g = torch.Generator().manual_seed(7 * tq + dh)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 1.2).to(torch.bfloat16)  # noqa: E731
    q, k, v = rnd(b, tq, h * dh), rnd(b, tk, h * dh), rnd(b, tk, h * dh)
    keep = None
    if masked:
        keep = torch.rand(b, h, tq, tk, generator=g) > 0.3
        keep[..., 0] = True  # (a row without any key is NaN in the reference too)
    sc = 1.0 / math.sqrt(dh) if scale is None else scale
    leaves, want_o, want_w = _weights_reference(q, k, v, h, dh, sc, keep, causal)
    g_o = torch.randn(b, tq, h * dh, generator=g)
    g_w = torch.randn(b, h, tq, tk, generator=g)
    ((want_o * g_o).sum() + (want_w * g_w).sum()).backward()

    dq, dk, dv = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    o, w = HF.attention_with_weights(dq, dk, dv, h, None if keep is None else keep.to(DEV), causal, dh, scale)
    assert w.dtype == torch.float32 and tuple(w.shape) == (b, h, tq, tk)
    assert_close(o, want_o.detach(), 1e-2, "output")
    assert_close(w, want_w.detach(), 2e-3, "weights")
    assert_close(w.sum(-1), torch.ones(b, h, tq), 1e-3, "rows sum to one")
    ((o.float() * g_o.to(DEV)).sum() + (w * g_w.to(DEV)).sum()).backward()
    for name, got, leaf in (("dq", dq.grad, leaves[0]), ("dk", dk.grad, leaves[1]), ("dv", dv.grad, leaves[2])):
        assert_close(got, leaf.grad, 2.5e-2, f"{name} (loss on output and weights)")
    # a loss on the weights alone: no output gradient exists, dv is None
    dq2, dk2, dv2 = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    _, w2 = HF.attention_with_weights(dq2, dk2, dv2, h, None if keep is None else keep.to(DEV), causal, dh, scale)
    (w2 * g_w.to(DEV)).sum().backward()
    l2 = [t.float().requires_grad_(True) for t in (q, k)]
    hd = lambda z, t: z.reshape(b, t, h, dh).permute(0, 2, 1, 3)  # noqa: E731
    raw = (hd(l2[0], tq) @ hd(l2[1], tk).transpose(-1, -2)) * sc
    drop = torch.zeros(b, h, tq, tk, dtype=torch.bool)
    if keep is not None:
        drop |= ~keep
    if causal:
        drop |= torch.ones(tq, tk, dtype=torch.bool).triu(1)
    (torch.softmax(raw.masked_fill(drop, float("-inf")), -1) * g_w).sum().backward()
    assert_close(dq2.grad, l2[0].grad, 1.5e-2, "dq (weights only)")
    assert_close(dk2.grad, l2[1].grad, 1.5e-2, "dk (weights only)")
    assert dv2.grad is None


def test_attention_module_returns_weights_like_the_reference_slow_path():
    """modules.Attention(require_weights=True) — self attention with a mask and `qk_scale` (the slow path is the one place where
    the reference applies it: attentions.py:262), and cross attention with separate k / v widths — against the reference's
    formula on the module's own parameters; `customize_sdp = True` takes the same path; overriding `_get_weights` is refused."""
    import cflearn_amd as C

    torch.manual_seed(3)
    att = C.modules.Attention(128, 2, is_self_attention=True, qk_scale=5.0).to(DEV).eval()
    x = torch.randn(2, 37, 128, device=DEV)
    mask = torch.rand(2, 37, 37, device=DEV) > 0.8  # True = zeroed (reference convention)
    mask[..., 0] = False
    out = att(x, x, x, mask=mask, require_weights=True)
    assert out.weights is not None and tuple(out.weights.shape) == (2, 2, 37, 37)
    bfr = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    qkv = bfr(torch.nn.functional.linear(bfr(x), bfr(att.in_w), att.qkv_bias))
    q, k, v = (t.reshape(2, 37, 2, 64).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    raw = (q @ k.transpose(-1, -2)).masked_fill(C.modules.expand_module_mask(mask, 2) == 0, float("-inf")) / 5.0
    w = torch.softmax(raw, -1)
    o = (w @ v).permute(0, 2, 1, 3).reshape(2, 37, 128)
    want = torch.nn.functional.linear(bfr(o), bfr(att.out_linear.linear.weight), att.out_linear.linear.bias)
    assert_close(out.weights, w, 3e-3, "module weights")
    assert_close(out.output, want, 1.5e-2, "module output")
    fast = att(x, x, x, mask=mask)  # the fused path ignores qk_scale, as the reference's sdp_attn call does
    assert fast.weights is None and not torch.allclose(fast.output.float(), out.output.float(), atol=1e-3)

    class Custom(C.modules.Attention):
        customize_sdp = True

    att2 = Custom(64, 1, k_dim=48, v_dim=32).to(DEV).eval()
    got = att2(torch.randn(2, 9, 64, device=DEV), torch.randn(2, 5, 48, device=DEV), torch.randn(2, 5, 32, device=DEV))
    assert tuple(got.weights.shape) == (2, 1, 9, 5)
    assert_close(got.weights.sum(-1), torch.ones(2, 1, 9), 1e-3, "cross-attention rows")

    class Sparse(C.modules.Attention):
        customize_sdp = True

        def _get_weights(self, raw_weights):
            return torch.relu(raw_weights)

    with pytest.raises(NotImplementedError):
        Sparse(64, 1, is_self_attention=True).to(DEV)(x[..., :64], x[..., :64], x[..., :64])


@pytest.mark.parametrize("dh,tq,tk,h", [(64, 37, 37, 2), (40, 130, 77, 2), (64, 300, 280, 1)])
def test_returned_weights_with_training_dropout(dh, tq, tk, h):
    """The reference's slow path in training mode (attentions.py:263-267): `weights = F.dropout(softmax)`, `output = weights @ v`,
    and the DROPPED weights are what is returned.  Here the output comes from the fused dropout kernels and the weights from
    cfhip_attn_probs times the same Philox keep-mask: checked against fp32 autograd given that mask (exported by
    cfhip_attn_dropout_mask for the (seed, offset) the call drew) — output, weights, and the gradients of a loss on BOTH."""
    from cflearn_amd import functional as HF

    b, p = 2, 0.3
    p_eff = ops.attn_dropout_p(p)
    assert p_eff == 77 / 256
    g = torch.Generator().manual_seed(dh + tq)
    q, k, v = (torch.randn(b, t, h * dh, generator=g).to(torch.bfloat16) for t in (tq, tk, tk))
    d_o = torch.randn(b, tq, h * dh, generator=g).to(torch.bfloat16)
    d_w = torch.randn(b, h, tq, tk, generator=g)
    ops.PhiloxState.manual_seed(99)
    seed, off0 = ops.PhiloxState.seed, ops.PhiloxState.offset
    leaves = [t.to(DEV).requires_grad_(True) for t in (q, k, v)]
    o, w = HF.attention_with_weights(*leaves, h, None, False, dh, None, p)
    assert ops.PhiloxState.offset - off0 == ops.attn_dropout_blocks(b, h, tq, tk)
    keep = ops.attn_dropout_mask(b, h, tq, tk, p, seed, off0).cpu()
    (o.float() * d_o.to(DEV).float()).sum().add((w * d_w.to(DEV)).sum()).backward()
    # fp32 reference with the same mask
    ref = [t.float().clone().requires_grad_(True) for t in (q, k, v)]
    hd = lambda t, n: t.reshape(b, n, h, dh).permute(0, 2, 1, 3)  # noqa: E731
    sm = torch.softmax(hd(ref[0], tq) @ hd(ref[1], tk).transpose(-1, -2) / math.sqrt(dh), -1)
    w_ref = sm * (keep.float() / (1.0 - p_eff))
    o_ref = (w_ref @ hd(ref[2], tk)).permute(0, 2, 1, 3).reshape(b, tq, h * dh)
    ((o_ref * d_o.float()).sum() + (w_ref * d_w).sum()).backward()
    assert torch.count_nonzero(w[(keep == 0).to(DEV)]) == 0
    assert_close(w, w_ref.detach(), 3e-3, "dropped weights")
    assert_close(o, o_ref.detach(), 1e-2, "output of the dropped weights")
    for name, got, want in zip("qkv", leaves, ref):
        assert_close(got.grad, want.grad, 2e-2, f"d{name} (loss on output + weights, dropout)")
    # eval mode / p = 0: no counters drawn, plain weights
    off1 = ops.PhiloxState.offset
    _, w0 = HF.attention_with_weights(*(t.detach() for t in leaves), h, None, False, dh, None, 0.0)
    assert ops.PhiloxState.offset == off1
    assert_close(w0.sum(-1), torch.ones(b, h, tq), 1e-3, "undropped rows sum to one")


def test_attention_module_returns_dropped_weights_in_training_mode():
    """modules.Attention(dropout > 0).train()(…, require_weights=True) no longer raises: the module's weights are
    zero exactly where its own mask dropped them, rows average to one over masks, and eval mode returns the plain softmax."""
    import cflearn_amd as C

    torch.manual_seed(4)
    att = C.modules.Attention(128, 2, is_self_attention=True, dropout=0.25).to(DEV)
    x = torch.randn(2, 41, 128, device=DEV)
    att.eval()
    w_eval = att(x, x, x, require_weights=True).weights
    assert_close(w_eval.sum(-1), torch.ones(2, 2, 41), 1e-3, "eval rows")
    att.train()
    ops.PhiloxState.manual_seed(7)
    seed, off0 = ops.PhiloxState.seed, ops.PhiloxState.offset
    xr = x.clone().requires_grad_(True)
    out = att(xr, xr, xr, require_weights=True)
    keep = ops.attn_dropout_mask(2, 2, 41, 41, 0.25, seed, off0)
    assert torch.count_nonzero(out.weights[keep == 0]) == 0
    assert_close(out.weights, w_eval * keep.float() / (1.0 - ops.attn_dropout_p(0.25)), 3e-3, "train weights = eval weights x mask / (1 - p)")
    (out.output.float().sum() + out.weights.square().sum()).backward()
    assert torch.isfinite(xr.grad).all() and all(p.grad is not None and torch.isfinite(p.grad).all() for p in att.parameters())
    ops.PhiloxState.manual_seed(7)
    again = att(x, x, x, require_weights=True)
    assert torch.equal(again.weights, out.weights) and torch.equal(again.output, out.output)


def test_returned_weights_at_a_unet_token_count_rows_sum_to_one_and_match_the_output():
    """size-independent properties of `attention_with_weights` at T = 4 096, head_dim 40 (a UNet level): every row of the
    returned weights sums to one, and weights @ v (fp32 on the host side of the check, a row sample) reproduces the fused
    kernels' output — the two results come from different kernels and must describe the same softmax."""
    from cflearn_amd import functional as HF

    b, t, h, dh = 1, 4096, 2, 40
    g = torch.Generator().manual_seed(11)
    q, k, v = ((torch.randn(b, t, h * dh, generator=g) * 0.8).to(torch.bfloat16).to(DEV) for _ in range(3))
    o, w = HF.attention_with_weights(q, k, v, h, None, False, dh, None)
    assert_close(w.sum(-1), torch.ones(b, h, t), 1e-3, "rows sum to one")
    rows = torch.randint(0, t, (64,), generator=g).to(DEV)
    vh = v.float().view(b, t, h, dh).permute(0, 2, 1, 3)            # [B, H, T, dh]
    want = (w[:, :, rows] @ vh).permute(0, 2, 1, 3).reshape(b, 64, h * dh)
    assert_close(o[:, rows], want, 1e-2, "weights @ v == fused output (sampled rows)")


@pytest.mark.parametrize("b,t,h", [(2, 197, 12), (1, 256, 3), (3, 130, 2), (40, 197, 12)])
def test_persistent_dq_pass_is_bit_identical_to_the_one_shot_kernel(b, t, h):
    """option "attn_persistent" bit 1 (round 4): the dQ pass as one 16-wave workgroup per CU walking (batch, head) pairs with the
    next head's K / V streaming in — the same instruction sequence per query tile as `attn_bwd_dq_kernel`, so dq (and the
    delta it leaves for the dK / dV pass, hence dk / dv) must be BIT-identical; 40 x 12 heads = more heads than workgroups."""
    qkv = _qkv(b, t, h, 300 + t, scale=1.3).to(DEV)
    d = h * 64
    d_o = torch.randn(b, t, d, generator=torch.Generator().manual_seed(t + 1)).to(torch.bfloat16).to(DEV)
    o, lse = ops.attn_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], h)
    outs = []
    try:
        for opt in (1, 3, 2, 0):
            ops.set_option("attn_persistent", opt)
            g = torch.zeros_like(qkv)
            ops.attn_bwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], o, d_o, lse, h,
                         dq=g[..., :d], dk=g[..., d:2 * d], dv=g[..., 2 * d:])
            outs.append(g)
    finally:
        ops.set_option("attn_persistent", 7)
    for g in outs[1:]:
        assert torch.equal(g, outs[0])
    if b <= 3:
        _, want = _oracle(qkv.cpu(), h, None, d_o.cpu())
        assert_close(outs[1], want, 2e-2, "persistent dq / dkv vs the oracle")


@pytest.mark.parametrize("b,t,h", [(2, 197, 12), (1, 256, 3), (3, 130, 2), (40, 197, 12)])
def test_persistent_forward_is_bit_identical_to_the_one_shot_kernel(b, t, h):
    """option "attn_persistent" bit 2: the forward as persistent 16-wave workgroups — same per-tile instruction sequence as
    `attn_fwd_kernel<NB, true>`: output and log-sum-exp bit-identical."""
    qkv = _qkv(b, t, h, 500 + t, scale=1.3).to(DEV)
    d = h * 64
    outs = []
    try:
        for opt in (1, 5, 7):
            ops.set_option("attn_persistent", opt)
            outs.append(ops.attn_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], h))
    finally:
        ops.set_option("attn_persistent", 7)
    for o, lse in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(lse, outs[0][1])
    if b <= 3:
        want_o, _ = _oracle(qkv.cpu(), h)
        assert_close(outs[1][0], want_o, 1e-2, "persistent forward vs the oracle")


@pytest.mark.parametrize("b,t,h", [(2, 197, 12), (1, 129, 3), (3, 160, 2), (2, 224, 4), (70, 197, 12), (3, 50, 12), (2, 33, 2), (2, 77, 8), (2, 96, 3),
                                   (2, 128, 2), (100, 50, 12)])
def test_one_pass_backward_matches_the_two_pass_kernels_and_the_oracle(b, t, h):
    """`attn_bwd_one_kernel` (round 6, option "attn_one_pass"): dQ, dK, dV of a head from ONE evaluation of S and dP (the dS tile goes
    through LDS once, in the bf16 rounding both passes of the two-pass form use) — against the two-pass kernels (same operands, same
    roundings: agreement to a few bf16 ulps of the accumulated sums) and, for the small cases, against fp32 math; `delta` is written
    as by the dQ pass; ragged last tiles (197 = 12 x 16 + 5), the largest supported length (224), and more heads than one round of
    persistent workgroups (70 x 12 = 840 heads on 256 workgroups: the walk over heads, the next head's Q / dO streaming in behind the
    last dQ phase); the short forms on 4 / 8 waves (T = 50: ViT-B/32 in CLIP; 33 .. 128); the 8-wave two-key-tile form of
    128 < T <= 224 (`attn_bwd_one2_kernel`, option value 2, the default) bit for bit against the 16-wave form."""
    qkv = _qkv(b, t, h, 300 + t, scale=1.5).to(DEV)
    d = h * 64
    d_o = torch.randn(b, t, d, generator=torch.Generator().manual_seed(t + 1)).to(torch.bfloat16).to(DEV)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    o, lse = ops.attn_fwd(q, k, v, h)
    outs = []
    for one in (0, 1, 1, 2):
        ops.set_option("attn_one_pass", one)
        try:
            dqkv = torch.zeros_like(qkv)
            delta = torch.zeros(b, h, t, device=DEV)
            ops.attn_bwd(q, k, v, o, d_o, lse, h, dq=dqkv[..., :d], dk=dqkv[..., d:2 * d], dv=dqkv[..., 2 * d:], delta=delta)
            outs.append((dqkv, delta))
        finally:
            ops.set_option("attn_one_pass", 2)
    torch.cuda.synchronize()
    assert torch.equal(outs[1][0], outs[2][0])  # run-to-run bitwise (no atomics)
    # option 2 (the default): 128 < T <= 224 on the 8-wave form with two key tiles per wave — the same sums in the same order
    assert torch.equal(outs[3][0], outs[1][0]) and torch.equal(outs[3][1], outs[1][1])
    for nm, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        assert_close(outs[1][0][..., sl], outs[0][0][..., sl].float(), 4e-3, f"one-pass {nm} vs two-pass {b}x{t}x{h}", abs_floor=1e-6)
    assert_close(outs[1][1], outs[0][1], 1e-5, "delta")
    if b <= 3:
        _, want_g = _oracle(qkv.cpu(), h, None, d_o.cpu())
        for nm, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
            assert_close(outs[1][0][..., sl], want_g[..., sl], 2e-2, f"one-pass {nm} vs fp32 {b}x{t}x{h}", abs_floor=1e-6)


@pytest.mark.parametrize("b,t,h", [(3, 77, 8), (2, 65, 2), (2, 128, 3), (2, 160, 2), (64, 77, 8)])
def test_one_pass_backward_causal(b, t, h):
    """the CAUSAL instantiations of `attn_bwd_one_kernel` (64 < T <= 160: the CLIP text tower's T = 77 with the triu(1) mask as a
    flag) against the two-pass kernels with the same flag and, for the small cases, against fp32 math with the explicit mask"""
    qkv = _qkv(b, t, h, 500 + t, scale=1.5).to(DEV)
    d = h * 64
    d_o = torch.randn(b, t, d, generator=torch.Generator().manual_seed(t + 2)).to(torch.bfloat16).to(DEV)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    o, lse = ops.attn_fwd(q, k, v, h, causal=True)
    outs = []
    for one in (0, 1):
        ops.set_option("attn_one_pass", one)
        try:
            dqkv = torch.zeros_like(qkv)
            ops.attn_bwd(q, k, v, o, d_o, lse, h, dq=dqkv[..., :d], dk=dqkv[..., d:2 * d], dv=dqkv[..., 2 * d:], causal=True)
            outs.append(dqkv)
        finally:
            ops.set_option("attn_one_pass", 2)
    for nm, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        assert_close(outs[1][..., sl], outs[0][..., sl].float(), 4e-3, f"causal one-pass {nm} vs two-pass {b}x{t}x{h}", abs_floor=1e-6)
    if b <= 3:
        keep = ~torch.triu(torch.ones(t, t, dtype=torch.bool), diagonal=1)
        _, want_g = _oracle(qkv.cpu(), h, keep, d_o.cpu())
        for nm, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
            assert_close(outs[1][..., sl], want_g[..., sl], 2e-2, f"causal one-pass {nm} vs fp32 {b}x{t}x{h}", abs_floor=1e-6)
