"""A12 (dropout / DropPath) and A16 (tabular encoder) on the GPU against the reference-made fixtures
tests/golden/stochastic.pt and tests/golden/ml_encoder.pt (oracle/gen_golden.py: gen_stochastic, gen_ml_encoder)."""
import pytest
import torch

import cflearn_amd as C
from cflearn_amd import functional as HF
from cflearn_amd import ops
from cflearn_amd.modules import CommonMLModule, DropPath, Dropout, MLEncoder
from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.mark.parametrize("name", ["f32", "bf16"])
def test_dropout_and_drop_path_bit_exact_given_the_mask(golden, name):
    g = golden("stochastic.pt")[name]
    x = g["x"].to(DEV)
    # nn.Dropout: the mask torch drew, injected -> bit-equal output; backward = the same mask on dy
    drop = Dropout(g["p"]).to(DEV)
    drop.inject_mask = g["mask"].to(DEV)
    xr = x.clone().requires_grad_(True)
    y = drop(xr)
    assert torch.equal(y.cpu(), g["y"])
    gy = torch.ones_like(y)
    y.backward(gy)
    want = (g["mask"].to(g["x"].dtype) / (1.0 - g["p"]))  # d/dx of x * noise
    assert torch.equal(xr.grad.cpu(), want)
    # DropPath (customs.py:434-443)
    dp = DropPath(g["rate"]).to(DEV)
    dp.inject_mask = g["mb"].to(DEV)
    xb = g["xb"].to(DEV).requires_grad_(True)
    yb = dp(xb)
    assert torch.equal(yb.cpu(), g["yb"])
    yb.backward(torch.ones_like(yb))
    want_b = (torch.ones_like(g["xb"]).div(1.0 - g["rate"]) * g["mb"].to(g["xb"].dtype).view(-1, 1, 1))
    assert torch.equal(xb.grad.cpu(), want_b)
    # eval mode / p == 0: identity (the same tensor object)
    drop.eval()
    assert drop(x) is x
    dp.eval()
    assert dp(xb) is xb


def test_philox_masks_statistics_and_reproducibility():
    n, p = 1 << 22, 0.3
    x = torch.ones(n, device=DEV)
    ops.PhiloxState.manual_seed(1234)
    y1 = HF.dropout(x, p, True)
    y2 = HF.dropout(x, p, True)           # the offset advanced: a different mask
    ops.PhiloxState.manual_seed(1234)
    y3 = HF.dropout(x, p, True)           # same seed, same offset: the same mask
    keep = (y1 != 0).float()
    assert abs(keep.mean().item() - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-4
    assert torch.equal(y1, y3) and not torch.equal(y1, y2)
    assert abs(((y1 != 0) & (y2 != 0)).float().mean().item() - (1 - p) ** 2) < 2e-3   # independent draws
    assert torch.allclose(y1[y1 != 0], torch.full((1,), 1 / (1 - p), device=DEV))
    # no mask tensor is stored: the backward regenerates it from (seed, offset)
    xr = torch.randn(1000, 33, device=DEV).to(torch.bfloat16).requires_grad_(True)
    yr = HF.dropout(xr, 0.5, True)
    yr.backward(torch.ones_like(yr))
    assert torch.equal(xr.grad != 0, yr != 0) or (xr.detach() == 0).any()
    # DropPath: per-sample, whole samples dropped, E[mask] = keep_prob
    ops.PhiloxState.manual_seed(7)
    xb = torch.ones(4096, 8, 16, device=DEV)
    yb = HF.drop_path(xb, 0.2, True)
    per = yb.reshape(4096, -1)
    assert ((per == 0).all(1) | (per == 1.25).all(1)).all()
    assert abs((per[:, 0] != 0).float().mean().item() - 0.8) < 4 * (0.16 / 4096) ** 0.5


def test_mixing_block_with_dropout_and_drop_path_matches_the_reference_formula(golden):
    """api.py:130-158 with both random masks injected: x + dp(drop(token_mix(LN(x)))), then the channel branch."""
    g = golden("vit_small.pt")
    cfg = dict(g["cfg"])
    m = C.build_module("cv_clf", config=dict(in_channels=3, num_classes=g["num_classes"], img_size=cfg.pop("img_size"),
                                             latent_dim=cfg["latent_dim"], encoder="vit",
                                             encoder_config=dict(cfg, dropout=0.25, drop_path_rate=0.5)))
    missing = m.load_state_dict(g["sd"])  # dropout has no parameters: the reference checkpoint loads unchanged
    assert not missing.missing_keys and not missing.unexpected_keys
    m = m.to(DEV)
    for b_ in m.encoder.encoder.mixing_blocks:
        # the same `dropout` also reaches Attention as the dropout ON THE ATTENTION PROBABILITIES (inside sdp_attn,
        # toolkit.py:953-963; tests/test_gpu_attn.py::test_attention_probability_dropout*): its mask comes from the
        # Philox stream and cannot be injected, so it is switched off for this formula check
        assert b_.token_mixing.net.dropout == 0.25
        b_.token_mixing.net.dropout = 0.0
    blk = m.encoder.encoder.mixing_blocks[1]
    assert blk.drop_path.dropout == 0.5 and blk.token_mixing_dropout.p == 0.25
    torch.manual_seed(0)
    x = torch.randn(4, 17, cfg["latent_dim"], device=DEV)
    m.train()
    # masks: token-mixing dropout (elementwise), DropPath on both branches (per sample), feed-forward dropouts
    dmask = (torch.rand(4, 17, cfg["latent_dim"], device=DEV) > 0.25).to(torch.uint8)
    pmask = torch.tensor([1.0, 0.0, 1.0, 1.0], device=DEV)
    blk.channel_mixing.dropout = 0.0  # isolate the block-level masks (FeedForward's own dropouts are tested below)
    blk.token_mixing_dropout.inject_mask = dmask
    blk.drop_path.inject_mask = pmask
    # first branch by hand, from the deterministic pieces
    with torch.no_grad():
        branch = blk.token_mixing(blk.token_norm(x)).float()
        want1 = x + (branch.to(torch.bfloat16) * (dmask.to(torch.bfloat16) / 0.75)).div(0.5).float() * pmask.view(-1, 1, 1)
    blk2_in = None

    def hook(mod, inp):
        nonlocal blk2_in
        blk2_in = inp[0].detach().clone()

    h = blk.channel_norm.register_forward_pre_hook(hook)
    y = blk(x)
    h.remove()
    assert_close(blk2_in, want1, 1e-6, "x + drop_path(dropout(token_mixing(LN(x))))", abs_floor=1e-6)
    assert y.shape == x.shape and torch.isfinite(y).all()
    # sample 1 was dropped on the first branch: its stream is untouched there
    assert torch.equal(blk2_in[1], x[1])
    # FeedForward's own dropouts (channel_mixers.py:30-41) with injected masks
    ff = blk.channel_mixing
    ff.dropout = 0.25
    h_in = torch.randn(4 * 17, cfg["latent_dim"], device=DEV).to(torch.bfloat16)
    m1 = (torch.rand(4 * 17, ff.latent_dim, device=DEV) > 0.25).to(torch.uint8)
    m2 = (torch.rand(4 * 17, cfg["latent_dim"], device=DEV) > 0.25).to(torch.uint8)
    ff.net[2].inject_mask, ff.net[4].inject_mask = m1, m2
    out = ff(h_in)
    with torch.no_grad():
        ff.eval()
        hidden = ff.net[0](h_in, act=HF.ACT_GELU)
        want = ff.net[3](hidden * (m1.to(torch.bfloat16) / 0.75)) * (m2.to(torch.bfloat16) / 0.75)
        ff.train()
    assert_close(out, want, 1e-6, "FeedForward with dropout", abs_floor=1e-6)
    # and the whole model trains with random masks (loss finite, gradients everywhere)
    logits = m(g["img"].to(DEV))["predictions"]
    loss = torch.nn.functional.cross_entropy(logits, g["labels"].view(-1).to(DEV))
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None for p in m.parameters())


@pytest.mark.parametrize("case", ["mixed", "all_embedding_oob"])
def test_ml_encoder_bit_exact(golden, case):
    g = golden("ml_encoder.pt")[case]
    enc = MLEncoder(g["settings"])
    assert list(enc.state_dict().keys()) == list(g["sd"].keys())
    enc.load_state_dict(g["sd"])
    assert enc.dim_increment == g["dim_increment"]
    enc = enc.to(DEV).eval()
    x = g["x"].to(DEV)
    indices, one_hot, embedding = enc.encode_result(x)
    assert torch.equal(indices.cpu(), g["indices"])                      # integer work: bit-exact
    if g["one_hot"] is None:
        assert one_hot is None
    else:
        assert torch.equal(one_hot.cpu(), g["one_hot"])
    assert torch.equal(embedding.cpu(), g["embedding"])                  # a gather of f32 rows: bit-exact
    merged = enc(x)
    assert torch.equal(merged.cpu(), g["merged_all"])
    # backward: scatter-add into the tables
    enc.train()
    for tab in enc.embeddings.values():
        tab.dropout = None
    enc.zero_grad()
    enc(x).backward(g["gy"].to(DEV))
    for k, p in enc.named_parameters():
        assert_close(p.grad, g["grads"][k], 1e-6, f"d {k}", abs_floor=1e-6)


def test_common_ml_module_trains_fcnn_on_categorical_data(golden):
    """models/ml/common.py:27-93: encoder -> FCNN, input_dim grown by the encoder's dim_increment; default embedding
    dropout 0.1 active in training (ml_encoder.py:113-117)."""
    from cflearn_amd.engine import TrainStep

    torch.manual_seed(0)
    settings = {"0": dict(dim=6, methods="one_hot"), "5": dict(dim=12, methods="embedding")}
    model = CommonMLModule("fcnn", dict(input_dim=8, output_dim=3, hidden_units=[64, 64]), settings).to(DEV)
    assert model.m["module"].net[0].linear.linear.weight.shape[1] == 8 + (6 - 1) + (4 - 1)
    b = 512
    x = torch.randn(b, 8)
    x[:, 0] = torch.randint(0, 6, (b,)).float()
    x[:, 5] = torch.randint(0, 12, (b,)).float()
    y = ((x[:, 0] + x[:, 5]).long() % 3)
    ts = TrainStep(model, lr=3e-3, weight_decay=0.0)
    ts.optimizer.lazy_zero = False  # the embedding tables receive their gradient through autograd accumulation
    xd, yd = x.to(DEV), y.to(DEV)
    losses = [ts.step(xd, yd).item() / b for _ in range(150)]
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
