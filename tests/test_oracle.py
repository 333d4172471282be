"""The CPU oracle (oracle/vit_oracle.py) against the golden fixtures frozen from the reference's own
modules (oracle/gen_golden.py), and — when /root/reference is present — against the reference live."""
import pytest
import torch

import vit_oracle as O
from helpers import max_abs

TOL = 2.0e-5


def test_linear(golden):
    g = golden("linear.pt")
    assert max_abs(O.linear(g["x"], g["sd"]["linear.weight"], g["sd"]["linear.bias"]), g["y"]) < TOL


def test_layernorm(golden):
    g = golden("layernorm.pt")
    assert g["eps"] == 1.0e-6  # NormFactory("layer") default (norms.py:118-119)
    assert max_abs(O.layer_norm(g["x"], g["w"], g["b"], g["eps"]), g["y"]) < TOL


def test_sdp(golden):
    g = golden("sdp.pt")
    assert max_abs(O.sdp_attention(g["q"], g["k"], g["v"]), g["y_nomask"]) < TOL
    assert max_abs(O.sdp_attention(g["q"], g["k"], g["v"], g["keep"]), g["y_causal"]) < TOL


def test_attention_mask_quirk(golden):
    g = golden("attention.pt")
    for tag, mk in (("r_nomask", None), ("r_mask", g["mask"])):
        y = O.self_attention(g["x"], g["sd"], "", g["heads"], mk)
        assert max_abs(y, g[tag]["y"]) < TOL, tag


def test_attention_grads(golden):
    g = golden("attention.pt")
    sd = {k: v.clone().requires_grad_(True) for k, v in g["sd"].items()}
    x = g["x"].clone().requires_grad_(True)
    y = O.self_attention(x, sd, "", g["heads"], g["mask"])
    y.backward(g["r_mask"]["gy"])
    assert max_abs(x.grad, g["r_mask"]["gx"]) < TOL
    for k, v in sd.items():
        assert max_abs(v.grad, g["r_mask"]["grads"][k]) < 5 * TOL, k


def test_feedforward(golden):
    g = golden("feedforward.pt")
    assert max_abs(O.feed_forward(g["x"], g["sd"], ""), g["y"]) < TOL


def test_vit_classifier(golden):
    g = golden("vit_small.pt")
    loss, logits, grads = O.loss_and_grads(g["img"], g["labels"], g["sd"], g["heads"], g["cfg"]["num_layers"])
    assert max_abs(logits, g["logits"]) < TOL
    assert abs(loss.item() - g["loss"].item()) < TOL
    assert set(grads) == set(g["grads"])
    for k in grads:
        assert max_abs(grads[k], g["grads"][k]) < 5 * TOL, k


def test_losses_known_values():
    logits = torch.tensor([[2.0, 0.0, 0.0], [0.0, 0.0, 3.0]])
    labels = torch.tensor([[0], [1]])
    ce = torch.nn.functional.cross_entropy(logits, labels.view(-1))
    assert abs(O.cross_entropy(logits, labels).item() - ce.item()) < 1e-6
    p = torch.softmax(logits, 1) + 1e-6
    py = p.gather(1, labels).squeeze(1)
    assert abs(O.focal_loss(logits, labels).item() - (-(py.log()) * (1 - py) ** 2).mean().item()) < 1e-6


def test_adamw_step_matches_torch():
    torch.manual_seed(0)
    p0 = torch.randn(257)
    for decoupled, cls in ((True, torch.optim.AdamW), (False, torch.optim.Adam)):
        p = torch.nn.Parameter(p0.clone())
        opt = cls([p], lr=1e-2, weight_decay=0.1)
        q, m, v = p0.clone(), torch.zeros(257), torch.zeros(257)
        for t in range(1, 4):
            g = torch.randn(257, generator=torch.Generator().manual_seed(t))
            p.grad = g.clone()
            opt.step()
            O.adamw_step(q, g, m, v, t, 1e-2, weight_decay=0.1, decoupled=decoupled)
        assert max_abs(q, p.detach()) < 1e-6


def test_oracle_vs_reference_live():
    """Build container only: run the reference modules themselves next to the oracle."""
    from refharness import load_reference, reference_available

    if not reference_available():
        pytest.skip("/root/reference not present (GPU box)")
    ref = load_reference()
    torch.manual_seed(3)
    enc = ref.ViTEncoder(img_size=32, patch_size=16, in_channels=3, latent_dim=64, num_layers=1)
    head = ref.Linear(64, 5)
    sd = {f"encoder.{k}": v.detach() for k, v in enc.state_dict().items()}
    sd.update({f"head.{k}": v.detach() for k, v in head.state_dict().items()})
    img = torch.randn(3, 3, 32, 32)
    with torch.no_grad():
        want = head(enc(img))
    assert max_abs(O.vit_classifier(img, sd, 1, 1), want) < TOL
    # the reference's own known-answer test for Attention (tests/test_blocks.py:147-176):
    # reference Attention == nn.MultiheadAttention with injected weights; here oracle == reference
    att = ref.Attention(64, 1, is_self_attention=True)
    x = torch.randn(2, 9, 64)
    mask = torch.rand(2, 9, 9) < 0.2
    mask[:, range(9), range(9)] = False
    asd = {k: v.detach() for k, v in att.state_dict().items()}
    with torch.no_grad():
        want = att(x, x, x, mask=mask).output
    assert max_abs(O.self_attention(x, asd, "", 1, mask), want) < TOL
