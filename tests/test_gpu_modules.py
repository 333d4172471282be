"""Drop-in modules on the MI355X vs (a) the golden fixtures frozen from the reference's own
modules and (b) the CPU oracle.  Weights are always COPIED from the fixture / oracle state_dict
(never RNG-matched), as SURVEY §8a' prescribes.

Tolerances (bf16 activations & GEMM operands, fp32 accumulate / stats / param grads):
outputs rel-L2 <= 1e-2, gradients rel-L2 <= 3e-2, loss |rel| <= 3e-3 on the small perturbed model
and <= 1e-3 on ViT-B/16 at init (the north-star bound)."""
import math

import pytest
import torch

import vit_oracle as O
from helpers import assert_close, rel_l2

pytestmark = pytest.mark.gpu

import cflearn_amd as C  # noqa: E402

DEV = "cuda"


def _grads(module):
    return {k: p.grad.detach().float().cpu() for k, p in module.named_parameters() if p.grad is not None}


def test_linear_golden(golden):
    g = golden("linear.pt")
    m = C.Linear(96, 40).to(DEV)
    m.load_state_dict(g["sd"])
    x = g["x"].to(DEV).requires_grad_(True)
    y = m(x)
    assert y.dtype == torch.bfloat16
    assert_close(y, g["y"], 1e-2, "linear y")
    y.backward(g["gy"].to(DEV).to(torch.bfloat16))
    assert_close(x.grad, g["gx"], 1.5e-2, "linear gx")
    assert_close(m.linear.weight.grad, g["gw"], 1.5e-2, "linear gw")
    assert_close(m.linear.bias.grad, g["gb"], 1e-2, "linear gb")
    assert m.linear.weight.grad.dtype == torch.float32


def test_layernorm_golden(golden):
    g = golden("layernorm.pt")
    m = C.NormFactory("layer").make(128).to(DEV)
    assert m.eps == g["eps"]
    with torch.no_grad():
        m.weight.copy_(g["w"]); m.bias.copy_(g["b"])
    x = g["x"].to(DEV).requires_grad_(True)
    y = m(x)
    assert_close(y, g["y"], 1e-2, "ln y")
    y.backward(g["gy"].to(DEV).to(torch.bfloat16))
    assert_close(x.grad, g["gx"], 2e-2, "ln gx")
    assert_close(m.weight.grad, g["gw"], 1e-2, "ln gw")
    assert_close(m.bias.grad, g["gb"], 1e-2, "ln gb")


def test_layernorm_4d_golden(golden):
    """`NormFactory("layer_norm")` on [B, C, H, W] — the reference's own per-sample `LN` (norms.py:30-46: unbiased std over C*H*W, eps
    added to it), cfhip_layernorm4d_fwd / _bwd — against vectors generated from the reference's class (oracle/gen_golden.py
    gen_layernorm4d): batch 3, its batch-1 branch, and the non-affine form; NCHW and channels_last inputs; bit-reproducible."""
    for g in golden("layernorm4d.pt"):
        affine = g["w"] is not None
        c = g["x"].shape[1]
        m = C.NormFactory("layer_norm").make(c, elementwise_affine=affine).to(DEV)
        assert type(m).__name__ == "LN" and m.eps == g["eps"] and sorted(m.state_dict()) == (["bias", "weight"] if affine else [])
        if affine:
            with torch.no_grad():
                m.weight.copy_(g["w"]); m.bias.copy_(g["b"])
        x = g["x"].to(DEV).requires_grad_(True)
        y = m(x)
        assert y.dtype == torch.bfloat16 and y.shape == x.shape
        assert_close(y, g["y"], 6e-3, "ln4d y")
        y.backward(g["gy"].to(DEV).to(torch.bfloat16))
        assert_close(x.grad, g["gx"], 2e-2, "ln4d gx")
        if affine:
            assert_close(m.weight.grad, g["gw"], 1e-2, "ln4d gw")
            assert_close(m.bias.grad, g["gb"], 1e-2, "ln4d gb")
        x2 = g["x"].to(DEV).to(memory_format=torch.channels_last)
        with torch.no_grad():
            assert torch.equal(m(x2), y.detach()) and torch.equal(m(g["x"].to(DEV)), y.detach())
    # a 3-D input of the same module is the plain last-dim LayerNorm (norms.py:32-33)
    m = C.NormFactory("layer_norm").make(64).to(DEV)
    t = torch.randn(2, 5, 64, device=DEV)
    assert_close(m(t), torch.nn.functional.layer_norm(t, (64,), m.weight, m.bias, m.eps), 6e-3, "LN on tokens")


def test_layernorm_4d_at_a_unet_level_vs_oracle():
    """[2, 320, 32, 32] (655 360 elements per sample: 40 slices) with an offset far from zero — the sums run in double across slices — against
    the fp64 evaluation of the oracle's formula, forward and all three gradients"""
    import conv_oracle as CO
    from cflearn_amd import functional as HF

    g = torch.Generator().manual_seed(8)
    x = (torch.randn(2, 320, 32, 32, generator=g) * 0.7 + 3.0).to(torch.bfloat16)
    w, b = torch.randn(320, generator=g) * 0.3 + 1.0, torch.randn(320, generator=g) * 0.3
    gy = torch.randn(2, 320, 32, 32, generator=g).to(torch.bfloat16)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = CO.layer_norm_4d(xr, wr, br, 1.0e-6)
    yr.backward(gy.double())
    xd = x.to(DEV).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = HF.layer_norm_4d(xd, wd, bd, 1.0e-6)
    y.backward(gy.to(DEV))
    assert_close(y, yr.detach().float(), 4e-3, "ln4d y")
    assert_close(xd.grad, xr.grad.float(), 6e-3, "ln4d gx")
    assert_close(wd.grad, wr.grad.float(), 2e-3, "ln4d gw")
    assert_close(bd.grad, br.grad.float(), 1e-4, "ln4d gb")


@pytest.mark.parametrize("tag", ["r_nomask", "r_mask"])
def test_attention_golden(golden, tag):
    """incl. the reference's 3-D mask layout quirk (attentions.py:246-253)."""
    g = golden("attention.pt")
    m = C.Attention(128, g["heads"], is_self_attention=True).to(DEV)
    m.load_state_dict(g["sd"])
    x = g["x"].to(DEV).requires_grad_(True)
    mask = g["mask"].to(DEV) if tag == "r_mask" else None
    out = m(x, x, x, mask=mask)
    assert out.weights is None
    assert_close(out.output, g[tag]["y"], 1.5e-2, "attention y")
    out.output.backward(g[tag]["gy"].to(DEV).to(torch.bfloat16))
    assert_close(x.grad, g[tag]["gx"], 3e-2, "attention gx")
    for k, v in _grads(m).items():
        assert_close(v, g[tag]["grads"][k], 3e-2, f"attention grad {k}")


@pytest.mark.parametrize("dim,heads", [(192, 2), (160, 4), (256, 8)])
def test_attention_module_other_head_dims(dim, heads):
    """`Attention` with head_dim != 64 (96 / 40 / 32) dispatches to the general-head_dim kernels instead of raising
    (VERDICT r1 #13); checked against the oracle restatement of attentions.py:198-279 incl. a 3-D mask."""
    torch.manual_seed(dim + heads)
    m = C.Attention(dim, heads, is_self_attention=True)
    with torch.no_grad():
        m.qkv_bias.normal_(0.0, 0.1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(3, 21, dim)
    mask = torch.rand(3, 21, 21) < 0.3
    mask[:, :, 0] = False  # no fully masked row
    for msk in (None, mask):
        xr = x.clone().requires_grad_(True)
        want = O.self_attention(xr, sd, "", heads, msk)
        gy = torch.randn_like(want)
        want.backward(gy)
        md = m.to(DEV)
        md.zero_grad()
        xd = x.to(DEV).requires_grad_(True)
        out = md(xd, xd, xd, mask=None if msk is None else msk.to(DEV)).output
        assert_close(out, want, 1.5e-2, f"attention dh={dim // heads} y")
        out.backward(gy.to(DEV).to(torch.bfloat16))
        assert_close(xd.grad, xr.grad, 3e-2, f"attention dh={dim // heads} gx")


def test_feedforward_golden(golden):
    g = golden("feedforward.pt")
    m = C.FeedForward(128, 256, 0.0).to(DEV)
    m.load_state_dict(g["sd"])
    x = g["x"].to(DEV).requires_grad_(True)
    y = m(x)
    assert_close(y, g["y"], 1e-2, "ff y")
    y.backward(g["gy"].to(DEV).to(torch.bfloat16))
    assert_close(x.grad, g["gx"], 2e-2, "ff gx")
    for k, v in _grads(m).items():
        assert_close(v, g["grads"][k], 2e-2, f"ff grad {k}")


def _small_vit(g):
    cfg = dict(g["cfg"])
    m = C.build_module("cv_clf", config=dict(in_channels=3, num_classes=g["num_classes"], img_size=cfg.pop("img_size"),
                                             latent_dim=cfg["latent_dim"], encoder="vit", encoder_config=cfg))
    m.load_state_dict(g["sd"])
    return m.to(DEV)


@pytest.mark.parametrize("fused", [True, False])
def test_vit_small_golden(golden, fused):
    """head(ViTEncoder(x)) logits / CE loss / every parameter gradient vs the reference run."""
    g = golden("vit_small.pt")
    m = _small_vit(g)
    for blk in m.encoder.encoder.mixing_blocks:
        blk.use_fused = fused
    logits = m(g["img"].to(DEV))["predictions"]
    assert logits.dtype == torch.float32
    assert_close(logits, g["logits"], 1e-2, "logits")
    loss = torch.nn.functional.cross_entropy(logits, g["labels"].view(-1).to(DEV))
    assert abs(loss.item() - g["loss"].item()) <= 3e-3 * abs(g["loss"].item())
    loss.backward()
    grads = _grads(m)
    assert set(grads) == set(g["grads"])
    worst = 0.0
    for k, v in grads.items():
        worst = max(worst, assert_close(v, g["grads"][k], 3e-2, f"grad {k}", abs_floor=2e-4))
    print(f"fused={fused}: worst grad rel-L2 {worst:.3e}")


def test_fused_equals_composed(golden):
    g = golden("vit_small.pt")
    outs = []
    for fused in (True, False):
        m = _small_vit(g)
        for blk in m.encoder.encoder.mixing_blocks:
            blk.use_fused = fused
        logits = m(g["img"].to(DEV))["predictions"]
        logits.sum().backward()
        outs.append((logits.detach(), _grads(m)))
    assert_close(outs[0][0], outs[1][0], 2e-3, "fused vs composed logits")
    for k in outs[0][1]:
        assert_close(outs[0][1][k], outs[1][1][k], 1e-2, f"fused vs composed {k}", abs_floor=1e-4)


def test_stack_node_equals_block_nodes(golden):
    """one autograd node for the whole stack with a ONE-word bf16 gradient stream (fused.GRAD_STREAM_WORDS = 1, rounds 1-5) == one
    node per block: same kernels, the only difference is a lossless bf16 -> f32 -> bf16 round trip of the inter-block gradient.
    (The default two-word stream keeps the second word from block to block, which per-block nodes cannot: compared with the oracle
    in test_two_word_gradient_stream_through_a_block_stack.)"""
    from cflearn_amd import fused

    g = golden("vit_small.pt")
    outs = []
    keep = fused.GRAD_STREAM_WORDS
    fused.GRAD_STREAM_WORDS = 1
    try:
        for stack in (True, False):
            m = _small_vit(g)
            m.encoder.encoder.fuse_stack = stack
            assert len(m.encoder.encoder.mixing_blocks) > 1
            logits = m(g["img"].to(DEV))["predictions"]
            torch.nn.functional.cross_entropy(logits, g["labels"].view(-1).to(DEV)).backward()
            outs.append((logits.detach(), _grads(m)))
    finally:
        fused.GRAD_STREAM_WORDS = keep
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        if "norm" in k:  # dgamma / dbeta fold a workgroup's waves through LDS float atomics: order not fixed
            assert_close(outs[0][1][k], outs[1][1][k], 1e-5, k)
        else:
            assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def test_hook_contract_and_lowrank():
    """IBasicHook before/after_forward (LoRA contract, hijacks.py:33-49) and the low-rank Linear."""

    class Hook(torch.nn.Module):
        def before_forward(self, inp, index=None):
            return inp * 2

        def after_forward(self, inp, out):
            return out + 1

    torch.manual_seed(0)
    m = C.HijackCustomLinear(64, 32, hook=Hook()).to(DEV)
    assert m.kwargs == {} and m.args == (64, 32)
    x = torch.randn(10, 64, device=DEV)
    y = m(x)
    want = torch.nn.functional.linear(x, m.linear.weight, m.linear.bias) + 1
    assert_close(y, want, 1e-2, "hooked linear")
    lr = C.Linear(64, 32, rank=8).to(DEV)
    with torch.no_grad():
        lr.w1.normal_(); lr.w2.normal_()
    want = torch.nn.functional.linear(torch.nn.functional.linear(x, lr.w1), lr.w2, lr.b)
    assert_close(lr(x), want, 1.5e-2, "low-rank linear")


def test_vit_b16_loss_within_1e3_of_cpu_reference():
    """North-star bound: ViT-B/16 224^2 vs the fp32 CPU reference on the same inputs and the same
    (reference-initialised) weights.
      * loss within 1e-3 relative (the north-star number);
      * logits: at least as close to fp32 as the reference's OWN bf16-autocast CPU run is, and <= 8e-3 rel-L2
        (measured on the MI355X: ours 7.25e-3, the reference's bf16-autocast run 8.00e-3 — at initialisation the
        logits are tiny sums of 0.02-scale weights, so bf16 operand rounding is a large fraction of them: 1e-3 on raw
        logits is not attainable by any bf16 implementation, the reference's included; the small perturbed model of
        smoke() sits at 2.8e-3); vs the bf16-autocast run itself <= 1.2e-2;
      * every parameter gradient of the full-size model vs the oracle's fp32 backward."""
    torch.manual_seed(0)
    m = C.vit_b16_classifier(num_classes=1000)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    img = torch.randn(2, 3, 224, 224, generator=g)
    labels = torch.randint(0, 1000, (2, 1), generator=g)
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    want_loss_t, want, want_grads = O.loss_and_grads(img, labels, sd, 12, 12)
    want_loss = want_loss_t.item()
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        want_bf16 = O.vit_classifier(img, sd, 12, 12).float()
    m = m.to(DEV)
    logits = m(img.to(DEV))["predictions"]
    loss, dlogits = C.ops.softmax_xent(logits, labels.to(DEV), 0.5)
    got_loss = loss.item() / 2
    rel_loss = abs(got_loss - want_loss) / abs(want_loss)
    ours_vs_f32 = rel_l2(logits, want)
    cpu16_vs_f32 = rel_l2(want_bf16, want)
    ours_vs_cpu16 = rel_l2(logits, want_bf16)
    print(f"ViT-B/16: loss rel err {rel_loss:.2e}; logits rel-L2 ours vs fp32 {ours_vs_f32:.2e}, reference bf16-autocast "
          f"vs fp32 {cpu16_vs_f32:.2e}, ours vs bf16-autocast {ours_vs_cpu16:.2e}")
    assert rel_loss <= 1e-3, (got_loss, want_loss)
    assert ours_vs_f32 <= min(8e-3, cpu16_vs_f32), (ours_vs_f32, cpu16_vs_f32)
    assert ours_vs_cpu16 <= 1.2e-2, ours_vs_cpu16
    # full-size gradient parity: every one of the 152 parameter tensors
    logits.backward(dlogits)
    worst = ("", 0.0)
    for k, v in _grads(m).items():
        err = assert_close(v, want_grads[k], 2.5e-2, f"ViT-B/16 grad {k}", abs_floor=1e-5)  # measured worst 1.3e-2
        if want_grads[k].abs().max() > 1e-5 and err > worst[1]:
            worst = (k, err)
    print(f"ViT-B/16: worst parameter-gradient rel-L2 {worst[1]:.2e} ({worst[0]})")


def test_vit_long_sequence_matches_oracle():
    """A ViT whose token count exceeds the LDS-resident attention limit (img 288, patch 16 -> 18 x 18 + 1 = 325
    tokens): logits, loss and every gradient vs the fp32 CPU oracle (oracle/vit_oracle.py, pinned against the
    reference).  Exercises the chunked attention kernels inside the fused block stack."""
    import vit_oracle as O

    torch.manual_seed(3)
    cfg = dict(patch_size=16, latent_dim=128, num_layers=2)
    m = C.VanillaClassifier(3, 10, 288, 128, encoder="vit", encoder_config=cfg)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    img = torch.randn(2, 3, 288, 288, generator=g)
    labels = torch.randint(0, 10, (2, 1), generator=g)
    want_loss, want_logits, want_grads = O.loss_and_grads(img, labels, sd, num_heads=2, num_layers=2)
    m = m.to(DEV)
    logits = m(img.to(DEV))["predictions"]
    assert_close(logits, want_logits, 1.5e-2, "long-sequence logits")
    loss = torch.nn.functional.cross_entropy(logits, labels.view(-1).to(DEV))
    assert abs(loss.item() - want_loss.item()) <= 3e-3 * abs(want_loss.item())
    loss.backward()
    for k, v in _grads(m).items():
        assert_close(v, want_grads[k], 4e-2, f"long-sequence grad {k}", abs_floor=2e-4)


def test_post_norm_stack_and_positional_interpolation_golden(golden):
    """VERDICT r1 'smaller unbuilt pieces': the post-norm `MixingBlock` / `MixedStackedEncoder` (api.py:160-185, no
    head norm, sequence positional encoding incl. its head-token offset quirk) and the ViT encoder at a NON-native
    resolution (bicubic positional-encoding interpolation, api.py:231-267) against reference-made fixtures."""
    g = golden("postnorm_interp.pt")
    pn = g["post_norm"]
    m = C.MixedStackedEncoder(**pn["cfg"])
    assert list(m.state_dict().keys()) == list(pn["sd"].keys())
    m.load_state_dict(pn["sd"])
    m = m.to(DEV)
    x = pn["x"].to(DEV).requires_grad_(True)
    y = m(x)
    assert_close(y, pn["y"], 1.5e-2, "post-norm encoder y")
    y.backward(pn["gy"].to(DEV).to(y.dtype))
    assert_close(x.grad, pn["gx"], 4e-2, "post-norm encoder gx")
    for k, v in _grads(m).items():
        assert_close(v, pn["grads"][k], 5e-2, f"post-norm grad {k}", abs_floor=2e-3)
    it = g["interp"]
    v = C.build_module("encoders.vit", config=dict(it["cfg"]))
    assert list(v.state_dict().keys()) == list(it["sd"].keys())
    v.load_state_dict(it["sd"])
    v = v.to(DEV)
    out = v(it["img"].to(DEV), hwp=tuple(it["hwp"]))
    assert_close(out, it["y"], 1.5e-2, "ViT at 48x32 (native 32x32)")
    out.backward(it["gy"].to(DEV).to(out.dtype))
    assert_close(v.encoder.pos_encoding.pos_encoding.grad, it["gpos"], 4e-2, "d pos_encoding through the bicubic resample",
                 abs_floor=1e-3)
    with pytest.raises(ValueError):
        v(it["img"].to(DEV))  # like upstream: a non-native resolution needs hwp


def test_forward_slices_on_streams_are_bit_equal_to_one_pass(golden):
    """`fused.FWD_HALVES`: the block stack's forward as 2 / 3 batch-slice pipelines on separate streams gives the same
    bits as one pass for the logits, and the same gradients (round 3: the backward is sliced too, `fused.BWD_HALVES`;
    its LayerNorm parameter gradients then add the slices in order)"""
    from cflearn_amd import fused

    g = golden("vit_small.pt")
    m = _small_vit(g)
    torch.manual_seed(3)
    x = torch.randn(9, *g["img"].shape[1:], device=DEV)
    keep = fused.FWD_HALVES
    try:
        res = {}
        for v in (1, 2, 3):
            fused.FWD_HALVES = v
            m.zero_grad(set_to_none=True)
            y = m(x)
            y = y["predictions"] if isinstance(y, dict) else y
            y.float().square().mean().backward()
            torch.cuda.synchronize()
            res[v] = (y.detach().clone(), [p.grad.detach().clone() for p in m.parameters()])
        for v in (2, 3):
            assert torch.equal(res[1][0], res[v][0]), v
            # (this fixture's width 128 takes the round-1 LayerNorm backward, whose dgamma / dbeta fold uses LDS float
            # atomics: run-to-run differences of an ulp; at ViT-B/16 size every gradient is bit-equal too —
            # tools/fwd_halves_ab.py, profiles/r02/fwd_halves_ab.log)
            for a, b in zip(res[1][1], res[v][1]):
                assert_close(b, a, 1e-5, f"gradient with {v} forward slices", abs_floor=1e-7)
    finally:
        fused.FWD_HALVES = keep


def test_sliced_passes_with_stale_weight_shadows(golden):
    """with forward slices on side streams, a STALE bf16 weight shadow (weights changed in place by a
    torch optimizer, no ParamArena; or the very first forward of a fresh module) used to be re-cast on the main stream
    AFTER the side stream had forked — the side slice could read old / half-written weights.  Fresh module, sliced from
    the first call, weights stepped in place between forwards; every step must match an identical model run as one pass
    on one stream."""
    import copy

    from cflearn_amd import fused

    g = golden("vit_small.pt")
    torch.manual_seed(5)
    xs = [torch.randn(8, *g["img"].shape[1:], device=DEV) for _ in range(4)]
    keep = fused.FWD_HALVES, fused.BWD_HALVES
    try:
        outs = {}
        for name, (fh, bh) in (("sliced", (2, 2)), ("one pass", (1, 1))):
            fused.FWD_HALVES, fused.BWD_HALVES = fh, bh
            m = _small_vit(g)  # fresh module: no shadow exists yet
            opt = torch.optim.SGD(m.parameters(), lr=0.05)
            rec = []
            for x in xs:
                opt.zero_grad(set_to_none=True)
                y = m(x)
                y = y["predictions"] if isinstance(y, dict) else y
                y.float().square().mean().backward()
                opt.step()  # in-place update: every shadow is stale at the next forward
                rec.append(y.detach().clone())
            torch.cuda.synchronize()
            outs[name] = rec
        for i, (a, b) in enumerate(zip(outs["sliced"], outs["one pass"])):
            assert_close(a, b, 2e-3 if i else 0.0, f"logits of step {i}", abs_floor=1e-6)
        assert torch.equal(outs["sliced"][0], outs["one pass"][0])
    finally:
        fused.FWD_HALVES, fused.BWD_HALVES = keep


def test_stand_alone_linear_weight_gradients_join_the_grouped_launches():
    """Round 3: the weight gradients of stand-alone Linear layers are queued (fused.queue_linear_dw) and written by grouped
    launches — flushed by tile count, at the end of the backward pass, or as split-K GEMMs when too few tiles are waiting.
    A chain of Linear layers incl. one SHARED weight (queued twice in one pass: the second problem must accumulate onto the
    first, never race it inside one launch), two backward passes without zeroing in between (accumulation onto existing
    `.grad`); compared with the on-the-spot path (LINEAR_DW_TILES = 0) and with fp32 torch."""
    from cflearn_amd import fused

    torch.manual_seed(11)
    dims = [512, 768, 768, 512, 1024, 512]
    x = torch.randn(384, dims[0], device=DEV)
    keep = fused.LINEAR_DW_TILES, fused.DW_MIN_TILES
    keep_rows, fused.LINEAR_DW_MIN_ROWS = fused.LINEAR_DW_MIN_ROWS, 0  # (the default queues only reductions of >= 49 152 rows: the 256^2 UNet's)
    try:
        res = {}
        for name, (tiles, floor) in (("grouped", (12, 1)), ("end of pass", (10 ** 6, 1)), ("few tiles", (10 ** 6, 10 ** 6)), ("on the spot", (0, 1))):
            fused.LINEAR_DW_TILES, fused.DW_MIN_TILES = tiles, floor
            torch.manual_seed(12)
            layers = [C.Linear(a, b).to(DEV) for a, b in zip(dims[:-1], dims[1:])]
            shared = layers[1]  # 768 -> 768, applied twice

            def run():
                h = x
                for i, l in enumerate(layers):
                    h = l(h)
                    if i == 1:
                        h = shared(h)
                return h.float().square().mean()

            run().backward()
            loss = run()
            loss.backward()  # second pass accumulates
            torch.cuda.synchronize()
            res[name] = [p.grad.detach().clone() for l in layers for p in l.parameters()]
            if name == "grouped":
                ws = [l.linear.weight.detach().double() for l in layers]
                bs = [l.linear.bias.detach().double() for l in layers]
        # fp32/64 torch reference of the same chain
        ps = [t.clone().requires_grad_(True) for pair in zip(ws, bs) for t in pair]
        h = x.double()
        for i in range(len(layers)):
            h = h @ ps[2 * i].t() + ps[2 * i + 1]
            if i == 1:
                h = h @ ps[2].t() + ps[3]
        (2.0 * h.square().mean()).backward()
        for name in ("grouped", "end of pass", "few tiles"):
            for i, (a, b) in enumerate(zip(res[name], res["on the spot"])):
                assert_close(a, b, 2e-3, f"{name}: gradient {i} vs the on-the-spot path", abs_floor=1e-7)
        for i, (a, r) in enumerate(zip(res["grouped"], ps)):
            assert_close(a, r.grad.float(), 3e-2, f"gradient {i} vs torch", abs_floor=1e-7)
        assert not fused._pending_dw
    finally:
        fused.LINEAR_DW_TILES, fused.DW_MIN_TILES = keep
        fused.LINEAR_DW_MIN_ROWS = keep_rows


def test_two_word_gradient_stream_through_a_block_stack():
    """fused.GRAD_STREAM_WORDS = 2 (what CLIP's towers ask for through their metas): the block stack carries the gradient of its f32
    residual stream in two bf16 words and hands an f32 gradient to what sits below it.  Against the fp32 oracle every parameter
    gradient is at least as close as with one word (same matrix products, fewer roundings on the residual path), the input-side
    tensors (head token, positional encoding, patch projection) strictly closer; the loss and the forward do not change; recorded
    and replayed launch plans give the same gradients bit for bit."""
    import vit_oracle as O
    from cflearn_amd import fused
    from cflearn_amd.engine import TrainStep

    torch.manual_seed(5)
    cfg = dict(patch_size=8, latent_dim=256, num_layers=6)
    g = torch.Generator().manual_seed(9)
    img = torch.randn(8, 3, 64, 64, generator=g)
    labels = torch.randint(0, 10, (8, 1), generator=g)
    m0 = C.VanillaClassifier(3, 10, 64, 256, encoder="vit", encoder_config=cfg)
    with torch.no_grad():
        for p in m0.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    sd = {k: v.detach().clone() for k, v in m0.state_dict().items()}
    want_loss, want_logits, want_grads = O.loss_and_grads(img, labels, sd, num_heads=4, num_layers=6)
    errs, logits_seen = {}, {}
    keep = fused.GRAD_STREAM_WORDS
    try:
        for words in (1, 2):
            fused.GRAD_STREAM_WORDS = words
            m = C.VanillaClassifier(3, 10, 64, 256, encoder="vit", encoder_config=cfg)
            m.load_state_dict(sd)
            m = m.to(DEV)
            ts = TrainStep(m, lr=0.0)
            per_step = []
            for _ in range(3):  # composed launch, plan recording, plan replay: lr = 0 keeps the problem fixed
                loss = ts.step(img.to(DEV), labels.view(-1).to(DEV)).item() / 8
                torch.cuda.synchronize()
                per_step.append({k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters()})
            assert abs(loss - want_loss.item()) <= 3e-3 * abs(want_loss.item())
            for k in per_step[0]:
                assert torch.equal(per_step[1][k], per_step[0][k]) and torch.equal(per_step[2][k], per_step[0][k]), (words, k)
            errs[words] = {k: rel_l2(v, want_grads[k]) for k, v in per_step[0].items()}
            logits_seen[words] = m(img.to(DEV))["predictions"].detach().clone()
    finally:
        fused.GRAD_STREAM_WORDS = keep
    assert torch.equal(logits_seen[1], logits_seen[2])
    worse = {k: (errs[1][k], errs[2][k]) for k in errs[1] if errs[2][k] > 1.1 * errs[1][k] + 1e-4}
    assert not worse, worse
    bottom = [k for k in errs[1] if "head_token" in k or "pos_encoding" in k or "to_patches" in k]
    assert bottom
    print({k: (round(errs[1][k], 5), round(errs[2][k], 5)) for k in bottom})
    assert all(errs[2][k] <= errs[1][k] for k in bottom)


@pytest.mark.parametrize("words", [1, 2], ids=["one_word_gradient_stream", "two_word_gradient_stream"])
def test_vit_b16_gradients_vs_the_reference_bf16_yardstick(words):
    """ViT-B/16 224^2 x 8 (the headline model) through `TrainStep`: all 152 parameter gradients against the fp32 oracle, bounded
    TENSOR BY TENSOR by the distance of the reference's own bf16-autocast run from its fp32 run (tests/golden/vit_b16_yardstick.pt,
    made from cflearn's ViTEncoder + Linear by oracle/gen_vit_b16_yardstick.py) instead of rounds 1-5's flat 2.5e-2; the same
    fixture pins oracle/vit_oracle.py at full size (loss, logits probe, gradient probes and norms of all 152 tensors)."""
    import os

    from gen_vit_b16_yardstick import BATCH, probe, seeded_problem

    from cflearn_amd import fused
    from cflearn_amd.engine import TrainStep

    ref = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vit_b16_yardstick.pt"), weights_only=False)
    sd, img, labels = seeded_problem()
    assert ref["batch"] == BATCH and ref["n_params"] == 86567656
    prev = torch.get_num_threads()
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    try:
        want_loss, want_logits, want_grads = O.loss_and_grads(img, labels, sd, 12, 12)
    finally:
        torch.set_num_threads(prev)
    assert abs(want_loss.item() - ref["loss_fp32"]) <= 1e-5 * ref["loss_fp32"]
    assert rel_l2(probe(want_logits, 64), ref["logits_probe"]) <= 1e-4
    for k, gr in want_grads.items():
        assert rel_l2(probe(gr), ref["grad_probe"][k]) <= 5e-4, k
        assert abs(gr.norm().item() - ref["grad_norm"][k]) <= 2e-4 * ref["grad_norm"][k], k
    keep = fused.GRAD_STREAM_WORDS
    fused.GRAD_STREAM_WORDS = words
    try:
        torch.manual_seed(0)
        m = C.vit_b16_classifier(num_classes=1000)
        m.load_state_dict(sd)
        m = m.to(DEV)
        ts = TrainStep(m, lr=0.0)
        for _ in range(3):  # composed launch, plan recording, plan replay (lr = 0: the same problem three times)
            loss = ts.step(img.to(DEV), labels.view(-1).to(DEV)).item() / BATCH
        torch.cuda.synchronize()
        errs = {k: rel_l2(p.grad, want_grads[k]) for k, p in m.named_parameters()}
    finally:
        fused.GRAD_STREAM_WORDS = keep
    assert abs(loss - want_loss.item()) <= 1e-3 * want_loss.item(), (loss, want_loss.item())
    ratio = {k: errs[k] / ref["grad_err"][k] for k in errs}
    order = sorted(ratio, key=ratio.get, reverse=True)
    print(f"ViT-B/16 x {BATCH}, {words}-word gradient stream: loss {loss:.6f} vs {want_loss.item():.6f}; gradients vs fp32 oracle: worst "
          f"{max(errs.values()):.3e}; against the reference's own bf16 distance: max x {ratio[order[0]]:.2f}, median x {ratio[order[len(order) // 2]]:.2f}")
    for k in order[:12]:
        print(f"    {k:74s} {errs[k]:.3e}   reference bf16-autocast {ref['grad_err'][k]:.3e} (x {ratio[k]:.2f})")
    # two words (the default): every tensor within 1.1 x the reference's own bf16 distance (measured max 1.09 x, median 0.96 x).
    # one word (rounds 1-5, CFHIP_GRAD_STREAM_WORDS=1): measured max 1.31 x, median 1.10 x — kept as a regression bound only
    bound = 1.1 if words == 2 else 1.45
    for k in errs:
        assert errs[k] <= max(5e-3, bound * ref["grad_err"][k]), (k, errs[k], ref["grad_err"][k])
