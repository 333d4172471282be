"""LayerNorm + element-wise / glue kernels vs fp32 torch references and the CPU oracle."""
import math

import pytest
import torch

import vit_oracle as O
from helpers import assert_close, bf16_round, max_abs

pytestmark = pytest.mark.gpu

from cflearn_amd import ops  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("m,d", [(12608, 768), (64, 768), (5, 128), (1000, 1024), (33, 2048), (17, 4), (300, 520)])
def test_layernorm_fwd_bwd(m, d):
    g = torch.Generator().manual_seed(m + d)
    x = (torch.randn(m, d, generator=g) * 2 + 0.5).to(torch.bfloat16)
    w = torch.randn(d, generator=g) * 0.2 + 1
    b = torch.randn(d, generator=g) * 0.2
    dy = torch.randn(m, d, generator=g).to(torch.bfloat16)
    xf = x.float().requires_grad_(True)
    wf, bf = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = O.layer_norm(xf, wf, bf, 1e-6)
    y_ref.backward(dy.float())
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6)
    assert_close(y, y_ref, 4e-3, "ln y")
    assert_close(mean, x.float().mean(1), 1e-5, "mean", abs_floor=1e-6)
    dx, dg, db = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd)
    assert_close(dx, xf.grad, 5e-3, "ln dx")
    assert_close(dg, wf.grad, 2e-4, "ln dgamma")
    assert_close(db, bf.grad, 2e-4, "ln dbeta")
    # fused residual-gradient add + accumulation into existing (adjacent) param grads
    add = torch.randn(m, d, generator=g).to(torch.bfloat16)
    pg = torch.ones(2 * d, device=DEV)
    dx2, _, _ = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, dx_add=add.to(DEV),
                                  dgamma=pg[:d], dbeta=pg[d:], accumulate=True)
    assert_close(dx2, xf.grad + add.float(), 5e-3, "ln dx + add")
    assert_close(pg[:d], wf.grad + 1, 2e-4, "dgamma acc")
    assert_close(pg[d:], bf.grad + 1, 2e-4, "dbeta acc")


@pytest.mark.parametrize("m,d", [(800, 768), (33, 2048), (300, 520), (17, 4)])
def test_layernorm_f32_rows_to_f32_rows(m, d):
    """A LayerNorm whose output is the residual stream (CLIP's `embedding_norm`, cv/encoder/transformer.py:60-64): f32 in, f32 out —
    against fp32 math to fp32 accuracy; the same statistics as the bf16-output launch; the autograd Function's backward; refused for bf16 rows."""
    from cflearn_amd import functional as HF

    g = torch.Generator().manual_seed(3 * m + d)
    x = torch.randn(m, d, generator=g) * 2 + 0.5
    w = torch.randn(d, generator=g) * 0.2 + 1
    b = torch.randn(d, generator=g) * 0.2
    xf, wf, bf = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    want = O.layer_norm(xf, wf, bf, 1e-5)
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, out_f32=True)
    assert y.dtype == torch.float32
    assert_close(y, want, 2e-6, "ln f32 y")
    y16, mean16, rstd16 = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5)
    assert y16.dtype == torch.bfloat16 and torch.equal(mean, mean16) and torch.equal(rstd, rstd16)
    assert torch.equal(y16, y.to(torch.bfloat16))  # the same arithmetic, one rounding apart
    with pytest.raises(ValueError):
        ops.layernorm_fwd(x.to(DEV).bfloat16(), w.to(DEV), b.to(DEV), 1e-5, out_f32=True)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    yd = HF.layer_norm(xd, wd, bd, 1e-5, True)
    assert yd.dtype == torch.float32 and torch.equal(yd.detach(), y)
    gy = torch.randn(m, d, generator=g).to(torch.bfloat16).float()
    want.backward(gy)
    yd.backward(gy.to(DEV))
    assert_close(xd.grad, xf.grad, 5e-3, "ln f32 dx")
    assert_close(wd.grad, wf.grad, 2e-4, "ln f32 dgamma")
    assert_close(bd.grad, bf.grad, 2e-4, "ln f32 dbeta")


def test_two_word_gradient_stream_split_join_and_layernorm_bwd2():
    """The residual-gradient stream in two bf16 words (round 6): f32 -> (hi, lo) is hi = bf16(v), lo = bf16(v - hi) bit for bit, the
    join hi + lo is within 2^-16 of v; `cfhip_layernorm_bwd2` adds BOTH words of dx_add in f32 and returns both words of dx: the
    pair is as close to the fp32 result as an f32 store would be (1e-5), the first word alone is the one-word kernel's result
    whenever no second word comes in, and the parameter gradients are those of the one-word call; the `_partials` form too."""
    g = torch.Generator().manual_seed(11)
    v = torch.randn(1000003, generator=g) * 3
    hi, lo = ops.split_f32(v.to(DEV))
    assert torch.equal(hi.cpu(), v.to(torch.bfloat16))
    assert torch.equal(lo.cpu(), (v - v.to(torch.bfloat16).float()).to(torch.bfloat16))
    back = ops.join_bf16x2(hi, lo)
    assert torch.equal(back.cpu(), hi.float().cpu() + lo.float().cpu())
    assert ((back.cpu() - v).abs() <= v.abs() * 2.0 ** -16).all()
    for m, d in ((800, 768), (1232, 512), (64, 320), (33, 2048), (17, 4)):
        x = torch.randn(m, d, generator=g) * 2 + 0.5
        w = torch.randn(d, generator=g) * 0.2 + 1
        b = torch.randn(d, generator=g) * 0.2
        dy = torch.randn(m, d, generator=g).to(torch.bfloat16)
        add = torch.randn(m, d, generator=g) * 4  # the incoming stream gradient, f32
        xf, wf, bf = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        O.layer_norm(xf, wf, bf, 1e-5).backward(dy.float())
        want = xf.grad + add
        xd, wd = x.to(DEV), w.to(DEV)
        _, mean, rstd = ops.layernorm_fwd(xd, wd, b.to(DEV), 1e-5)
        ahi, alo = ops.split_f32(add.to(DEV))
        dhi, dlo = torch.empty(m, d, dtype=torch.bfloat16, device=DEV), torch.empty(m, d, dtype=torch.bfloat16, device=DEV)
        dx, dg, db = ops.layernorm_bwd(dy.to(DEV), xd, wd, mean, rstd, dx_add=ahi, dx_add_lo=alo, dx_out=dhi, dx_lo_out=dlo)
        assert dx.data_ptr() == dhi.data_ptr()
        assert_close(ops.join_bf16x2(dhi, dlo), want, 2e-5, f"two-word dx {m}x{d}")
        assert_close(dg, wf.grad, 2e-4, "dgamma")
        assert_close(db, bf.grad, 2e-4, "dbeta")
        # one-word in, two words out == the one-word kernel on the first word (+ a residual); no second word out: the old result
        one, dg1, db1 = ops.layernorm_bwd(dy.to(DEV), xd, wd, mean, rstd, dx_add=ahi)
        two, _, _ = ops.layernorm_bwd(dy.to(DEV), xd, wd, mean, rstd, dx_add=ahi, dx_lo_out=torch.empty_like(dlo))
        assert torch.equal(one, two)
        assert_close(dg1, dg, 1e-6, "dgamma, two instantiations")  # (same sums; the compiler contracts the two kernels differently)
        assert_close(db1, db, 1e-6, "dbeta, two instantiations")
        # the `_partials` form (row kernel now, column reduce later)
        dhi2, dlo2 = torch.empty_like(dhi), torch.empty_like(dlo)
        _, ws, rows = ops.layernorm_bwd_partials(dy.to(DEV), xd, wd, mean, rstd, dx_add=ahi, dx_add_lo=alo, dx_out=dhi2, dx_lo_out=dlo2)
        assert torch.equal(dhi2, dhi) and torch.equal(dlo2, dlo)
        pg = torch.zeros(2 * d, device=DEV)
        ops.layernorm_bwd_reduce(ws, rows, d, pg[:d], pg[d:], False)
        assert_close(pg[:d], wf.grad, 2e-4, "partials dgamma")


def test_layernorm_strided_rows():
    """head LN reads token 0 of every sample: row stride T*D."""
    b, t, d = 16, 197, 768
    x = torch.randn(b, t, d).to(torch.bfloat16).to(DEV)
    w, bb = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
    y, _, _ = ops.layernorm_fwd(x[:, 0], w, bb, 1e-6)
    assert_close(y, O.layer_norm(x[:, 0].float().cpu(), w.cpu(), bb.cpu()), 4e-3, "strided ln")


def test_casts_gelu_add_transpose():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1000003, generator=g) * 3
    xb = ops.to_bf16(x.to(DEV))
    assert torch.equal(xb.cpu(), x.to(torch.bfloat16))  # round-to-nearest-even, bit exact
    assert torch.equal(ops.to_f32(xb).cpu(), x.to(torch.bfloat16).float())
    v = (torch.randn(4099, 33, generator=g) * 2).to(torch.bfloat16)
    assert_close(ops.gelu_fwd(v.to(DEV)), O.gelu_erf(v.float()), 4e-3, "gelu")
    vf = v.float().requires_grad_(True)
    dy = torch.randn(4099, 33, generator=g).to(torch.bfloat16)
    O.gelu_erf(vf).backward(dy.float())
    assert_close(ops.gelu_bwd(dy.to(DEV), v.to(DEV)), vf.grad, 4e-3, "gelu bwd")
    assert_close(ops.add(v.to(DEV), dy.to(DEV)), v.float() + dy.float(), 4e-3, "add")
    t = torch.randn(197, 130, generator=g).to(torch.bfloat16)
    assert torch.equal(ops.transpose(t.to(DEV)).cpu(), t.t().contiguous())


def test_im2row_and_patch_gemm_match_conv():
    g = torch.Generator().manual_seed(1)
    img = torch.randn(3, 3, 64, 64, generator=g)
    w = torch.randn(32, 3, 16, 16, generator=g) * 0.05
    bias = torch.randn(32, generator=g)
    rows = ops.im2row(img.to(DEV), 16)
    assert rows.shape == (3 * 16, 768)
    want_rows = img.reshape(3, 3, 4, 16, 4, 16).permute(0, 2, 4, 1, 3, 5).reshape(48, 768)
    assert torch.equal(rows.cpu(), want_rows.to(torch.bfloat16))
    out = ops.gemm(rows, w.reshape(32, -1).to(torch.bfloat16).to(DEV), bias=bias.to(DEV), out_dtype=torch.float32)
    conv = torch.nn.functional.conv2d(bf16_round(img), bf16_round(w), bias, stride=16)
    assert_close(out, conv.flatten(2).transpose(1, 2).reshape(48, 32), 1e-4, "patch embed == conv")
    rows_b = ops.im2row(img.to(torch.bfloat16).to(DEV), 16)
    assert torch.equal(rows_b.cpu(), rows.cpu())


def test_assemble_tokens():
    b, np_, d = 5, 16, 128
    g = torch.Generator().manual_seed(2)
    patches = torch.randn(b * np_, d, generator=g).to(torch.bfloat16)
    head, pos = torch.randn(d, generator=g), torch.randn((np_ + 1) * d, generator=g)
    x0 = ops.assemble_tokens_fwd(patches.to(DEV), head.to(DEV), pos.to(DEV), b)
    want = torch.cat([head.view(1, 1, d).expand(b, 1, d), patches.float().view(b, np_, d)], 1) + pos.view(1, np_ + 1, d)
    assert_close(x0, want, 4e-3, "assemble fwd")
    dx0 = torch.randn(b, np_ + 1, d, generator=g).to(torch.bfloat16)
    dhead, dpos = torch.zeros(d, device=DEV), torch.zeros((np_ + 1) * d, device=DEV)
    dp = ops.assemble_tokens_bwd(dx0.to(DEV), dhead, dpos, False)
    assert torch.equal(dp.cpu(), dx0[:, 1:].reshape(b * np_, d))
    assert_close(dpos, dx0.float().sum(0).reshape(-1), 1e-5, "dpos")
    assert_close(dhead, dx0.float()[:, 0].sum(0), 1e-5, "dhead")
    # round 6: the batch-parallel vector kernel (one workgroup per token and 256-column block, eight batch groups) at the ViT / CLIP step's
    # shapes, a width that is no multiple of 256, a batch that is no multiple of 8, accumulation onto existing gradients, the scalar form (D % 8 != 0)
    for b2, np2, d2 in ((128, 196, 768), (13, 7, 328), (3, 2, 20)):
        dx = torch.randn(b2, np2 + 1, d2, generator=g).to(torch.bfloat16)
        dh0, dp0 = torch.randn(d2, generator=g), torch.randn((np2 + 1) * d2, generator=g)
        dh, dq = dh0.to(DEV), dp0.to(DEV)
        out = ops.assemble_tokens_bwd(dx.to(DEV), dh, dq, True)
        assert torch.equal(out.cpu(), dx[:, 1:].reshape(b2 * np2, d2))
        assert_close(dq, dp0 + dx.float().sum(0).reshape(-1), 2e-6, f"dpos {b2}x{np2}x{d2} (+=)")
        assert_close(dh, dh0 + dx.float()[:, 0].sum(0), 2e-6, f"dhead {b2}x{np2}x{d2} (+=)")
        dq2 = torch.empty_like(dq)
        ops.assemble_tokens_bwd(dx.to(DEV), None, dq2, False, want_dpatches=False)
        dq3 = torch.empty_like(dq)
        ops.assemble_tokens_bwd(dx.to(DEV), None, dq3, False, want_dpatches=False)
        assert torch.equal(dq2, dq3)  # fixed summation order


def test_adam_matches_oracle():
    n = 100003
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    for decoupled in (False, True):
        p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        p16 = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
        q, mq, vq = p0.clone(), torch.zeros(n), torch.zeros(n)
        for t in range(1, 5):
            grad = torch.randn(n, generator=g)
            ops.adam_step(p, (grad * 2).to(DEV), m, v, p16, lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8,
                          weight_decay=0.05, decoupled=decoupled, step=t, grad_scale=0.5)
            O.adamw_step(q, grad, mq, vq, t, 1e-2, weight_decay=0.05, decoupled=decoupled)
        assert max_abs(p, q) < 2e-6
        assert torch.equal(p16.cpu(), p.cpu().to(torch.bfloat16))


def test_sumsq_and_xent():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1234567, generator=g)
    got = ops.sumsq(x.to(DEV)).item()
    assert abs(got - (x.double() ** 2).sum().item()) / got < 1e-5
    logits = torch.randn(64, 1000, generator=g) * 3
    labels = torch.randint(0, 1000, (64, 1), generator=g)
    loss, dl = ops.softmax_xent(logits.to(DEV), labels.to(DEV), 1.0 / 64)
    lf = logits.clone().requires_grad_(True)
    ref = O.cross_entropy(lf, labels)
    ref.backward()
    assert abs(loss.item() / 64 - ref.item()) < 1e-5 * abs(ref.item()) + 1e-6
    assert_close(dl, lf.grad, 1e-5, "dlogits")


def test_layernorm_bwd_split_modes():
    """dx-only and parameter-gradient-only launches give the same numbers as the combined one."""
    m, d = 1000, 768
    g = torch.Generator().manual_seed(5)
    x = torch.randn(m, d, generator=g).to(DEV)  # f32 residual stream
    dy = torch.randn(m, d, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(d, generator=g) * 0.2 + 1).to(DEV)
    b = torch.zeros(d, device=DEV)
    _, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-6)
    dx_all, dg_all, db_all = ops.layernorm_bwd(dy, x, w, mean, rstd, dx_add=dy)
    dx_only, none1, none2 = ops.layernorm_bwd(dy, x, w, mean, rstd, dx_add=dy, want_param_grads=False)
    assert none1 is None and none2 is None
    # (the combined launch is the half-wave-per-row kernel: other summation order than the dx-only kernel)
    assert_close(dx_only, dx_all, 2e-3, "dx split")
    pg = torch.zeros(2 * d, device=DEV)
    nodx, _, _ = ops.layernorm_bwd(dy, x, w, mean, rstd, dgamma=pg[:d], dbeta=pg[d:], want_dx=False)
    assert nodx is None
    assert_close(pg[:d], dg_all, 1e-5, "dgamma split")
    assert_close(pg[d:], db_all, 1e-5, "dbeta split")


@pytest.mark.parametrize("m,d,xf32", [(25216, 768, True), (1001, 768, False), (7, 256, True), (130, 512, False),
                                      (999, 1024, True), (513, 1280, True), (1, 768, True),
                                      # round 5: widths that are not multiples of 256 (the UNet's 320 / 640-wide token rows; a last group of
                                      # one lane; a single lane at all; the widest padded form)
                                      (4096, 320, False), (1000, 640, False), (77, 520, True), (9, 8, False), (513, 1272, False)])
def test_layernorm_bwd_one_launch(m, d, xf32):
    """The one-launch backward (dx + dgamma / dbeta, dy and x read once; reference norms.py:88-119 = nn.LayerNorm):
    vs torch autograd in fp32, vs the round-1 kernels (option ln_bwd_fused = 0), and bitwise run-to-run."""
    g = torch.Generator().manual_seed(m * 7 + d)
    x = torch.randn(m, d, generator=g) * 2 + 0.5
    if not xf32:
        x = x.to(torch.bfloat16)
    w = torch.randn(d, generator=g) * 0.2 + 1
    b = torch.randn(d, generator=g) * 0.2
    dy = torch.randn(m, d, generator=g).to(torch.bfloat16)
    add = torch.randn(m, d, generator=g).to(torch.bfloat16)
    xr = x.float().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-6).backward(dy.float())
    xd, wd, bd, dyd, addd = x.to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV), add.to(DEV)
    _, mean, rstd = ops.layernorm_fwd(xd, wd, bd, 1e-6)
    pg = torch.full((2 * d,), 0.5, device=DEV)
    dx, _, _ = ops.layernorm_bwd(dyd, xd, wd, mean, rstd, dx_add=addd, dgamma=pg[:d], dbeta=pg[d:], accumulate=True)
    assert_close(dx, xr.grad + add.float(), 5e-3, "dx + add")
    assert_close(pg[:d], wr.grad + 0.5, 2e-4, "dgamma (+=)", abs_floor=1e-4)
    assert_close(pg[d:], br.grad + 0.5, 2e-4, "dbeta (+=)", abs_floor=1e-4)
    # bitwise reproducible: no atomics anywhere
    pg2 = torch.full((2 * d,), 0.5, device=DEV)
    dx2, _, _ = ops.layernorm_bwd(dyd, xd, wd, mean, rstd, dx_add=addd, dgamma=pg2[:d], dbeta=pg2[d:], accumulate=True)
    assert torch.equal(dx, dx2) and torch.equal(pg, pg2)
    # the round-1 path on the same inputs
    ops.set_option("ln_bwd_fused", 0)
    try:
        dx0, dg0, db0 = ops.layernorm_bwd(dyd, xd, wd, mean, rstd, dx_add=addd)
    finally:
        ops.set_option("ln_bwd_fused", 1)
    assert_close(dx, dx0, 2e-3, "dx vs round-1 kernel")
    assert_close(pg[:d] - 0.5, dg0, 1e-4, "dgamma vs round-1 kernel", abs_floor=1e-4)
    assert_close(pg[d:] - 0.5, db0, 1e-4, "dbeta vs round-1 kernel", abs_floor=1e-4)
