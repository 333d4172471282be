"""Generate the golden fixtures under tests/golden/ from the REFERENCE'S OWN modules.

Run in the build container only (needs /root/reference):

    python oracle/gen_golden.py

For every fixture the reference module is built with a fixed seed, run in fp32 on the CPU on
seeded inputs, and {config, state_dict, inputs, outputs, grads} are frozen to a `.pt` file.
Before writing, the CPU restatement in `oracle/vit_oracle.py` is checked against the reference
output on the same inputs (max-abs tolerance printed and asserted), so a committed fixture is
also a record that the oracle was pinned when it was made.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import vit_oracle as O  # noqa: E402
import conv_oracle as CO  # noqa: E402
import clip_oracle as CL  # noqa: E402
import unet_oracle as UO  # noqa: E402
from refharness import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
ATOL = 2.0e-5


def _check(name: str, a: torch.Tensor, b: torch.Tensor, atol: float = ATOL) -> None:
    err = (a - b).abs().max().item()
    scale = b.abs().max().item()
    print(f"  oracle vs reference [{name}]: max|diff| = {err:.3e} (max|ref| = {scale:.3e})")
    assert err <= atol * max(1.0, scale), name


def gen_linear(ref) -> None:
    torch.manual_seed(11)
    m = ref.Linear(96, 40)
    with torch.no_grad():
        m.linear.bias.normal_()
    x = torch.randn(5, 7, 96, requires_grad=True)
    y = m(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    _check("linear", O.linear(x.detach(), sd["linear.weight"], sd["linear.bias"]), y.detach())
    torch.save(
        dict(
            sd=sd, x=x.detach(), y=y.detach(), gy=gy, gx=x.grad.clone(),
            gw=m.linear.weight.grad.clone(), gb=m.linear.bias.grad.clone(),
        ),
        os.path.join(OUT, "linear.pt"),
    )


def gen_layernorm(ref) -> None:
    torch.manual_seed(12)
    m = ref.NormFactory("layer").make(128)
    with torch.no_grad():
        m.weight.normal_(1.0, 0.2)
        m.bias.normal_(0.0, 0.2)
    x = (torch.randn(6, 9, 128) * 2.0 + 0.5).requires_grad_(True)
    y = m(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    _check("layer_norm", O.layer_norm(x.detach(), m.weight.detach(), m.bias.detach(), m.eps), y.detach())
    torch.save(
        dict(
            eps=m.eps, w=m.weight.detach().clone(), b=m.bias.detach().clone(), x=x.detach(),
            y=y.detach(), gy=gy, gx=x.grad.clone(), gw=m.weight.grad.clone(), gb=m.bias.grad.clone(),
        ),
        os.path.join(OUT, "layernorm.pt"),
    )


def gen_layernorm4d(ref) -> None:
    """the reference's own `LN` on [B, C, H, W] (norms.py:30-46), batch 3 and batch 1 (its `batch_size == 1` branch), affine and not"""
    torch.manual_seed(21)
    cases = []
    for (b, c, h, w), affine in (((3, 16, 9, 7), True), ((1, 32, 8, 8), True), ((2, 8, 5, 6), False)):
        m = ref.NormFactory("layer_norm").make(c, elementwise_affine=affine)
        assert type(m).__name__ == "LN" and m.eps == 1.0e-6
        if affine:
            with torch.no_grad():
                m.weight.normal_(1.0, 0.3)
                m.bias.normal_(0.0, 0.3)
        x = (torch.randn(b, c, h, w) * 1.7 + 0.4).requires_grad_(True)
        y = m(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        wt = m.weight.detach().clone() if affine else None
        bs = m.bias.detach().clone() if affine else None
        _check(f"layer_norm_4d {b}x{c}x{h}x{w}", CO.layer_norm_4d(x.detach(), wt, bs, m.eps), y.detach())
        cases.append(dict(eps=m.eps, w=wt, b=bs, x=x.detach().clone(), y=y.detach().clone(), gy=gy, gx=x.grad.clone(),
                          gw=m.weight.grad.clone() if affine else None, gb=m.bias.grad.clone() if affine else None))
    torch.save(cases, os.path.join(OUT, "layernorm4d.pt"))


def gen_attention(ref) -> None:
    """Self-attention with and without the 3-D bool mask (mask quirk, attentions.py:246-253)."""
    torch.manual_seed(13)
    heads = 2
    m = ref.Attention(128, heads, is_self_attention=True)
    with torch.no_grad():
        m.qkv_bias.normal_(0.0, 0.1)
        m.in_w.mul_(8.0)  # larger logits than the 0.02-std init: a peaky softmax
        m.out_linear.linear.bias.normal_(0.0, 0.1)
    b, t = 3, 21
    x = torch.randn(b, t, 128, requires_grad=True)
    mask = torch.rand(b, t, t) < 0.3
    mask[:, torch.arange(t), torch.arange(t)] = False  # no fully-masked rows
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    out = {}
    for tag, mk in (("r_nomask", None), ("r_mask", mask)):
        if x.grad is not None:
            x.grad = None
        m.zero_grad()
        y = m(x, x, x, mask=mk).output
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
        y.backward(gy)
        _check(f"attention/{tag}", O.self_attention(x.detach(), sd, "", heads, mk), y.detach())
        out[tag] = dict(
            y=y.detach().clone(), gy=gy, gx=x.grad.clone(),
            grads={k: p.grad.clone() for k, p in m.named_parameters()},
        )
    torch.save(dict(sd=sd, x=x.detach(), mask=mask, heads=heads, **out), os.path.join(OUT, "attention.pt"))


def gen_sdp(ref) -> None:
    """`sdp_attn` core on [B,H,T,dh] incl. a causal keep-mask (the CLIP text tower's use)."""
    torch.manual_seed(14)
    b, h, t, dh = 2, 3, 37, 64
    q, k, v = (torch.randn(b, h, t, dh) for _ in range(3))
    causal_keep = ~torch.triu(torch.ones(t, t, dtype=torch.bool), diagonal=1)
    y0 = ref.sdp_attn(q, k, v, False)
    y1 = ref.sdp_attn(q, k, v, False, causal_keep)
    _check("sdp/nomask", O.sdp_attention(q, k, v), y0)
    _check("sdp/causal", O.sdp_attention(q, k, v, causal_keep), y1)
    torch.save(dict(q=q, k=k, v=v, keep=causal_keep, y_nomask=y0, y_causal=y1), os.path.join(OUT, "sdp.pt"))


def gen_feedforward(ref) -> None:
    torch.manual_seed(15)
    m = ref.FeedForward(128, 256, 0.0)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.normal_(0.0, 0.1)
    x = torch.randn(4, 10, 128, requires_grad=True)
    y = m(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    _check("feed_forward", O.feed_forward(x.detach(), sd, ""), y.detach())
    torch.save(
        dict(sd=sd, x=x.detach(), y=y.detach(), gy=gy, gx=x.grad.clone(),
             grads={k: p.grad.clone() for k, p in m.named_parameters()}),
        os.path.join(OUT, "feedforward.pt"),
    )


def gen_vit(ref) -> None:
    """Small ViT classifier = head(ViTEncoder(x)), CE loss, all grads."""
    torch.manual_seed(16)
    cfg = dict(img_size=32, patch_size=8, in_channels=3, latent_dim=128, num_layers=2,
               feedforward_dim_ratio=2.0)
    num_classes, heads = 10, 2
    enc = ref.ViTEncoder(**cfg)
    head = ref.Linear(cfg["latent_dim"], num_classes)
    with torch.no_grad():  # perturb the zero / unit inits so every gradient path is exercised
        for n, p in list(enc.named_parameters()) + list(head.named_parameters()):
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
    b = 4
    img = torch.randn(b, 3, 32, 32)
    labels = torch.randint(0, num_classes, (b, 1))
    logits = head(enc(img))
    loss = torch.nn.functional.cross_entropy(logits, labels.view(-1))
    loss.backward()
    sd = {f"encoder.{k}": v.detach().clone() for k, v in enc.state_dict().items()}
    sd.update({f"head.{k}": v.detach().clone() for k, v in head.state_dict().items()})
    grads = {f"encoder.{k}": p.grad.clone() for k, p in enc.named_parameters()}
    grads.update({f"head.{k}": p.grad.clone() for k, p in head.named_parameters()})
    o_loss, o_logits, o_grads = O.loss_and_grads(img, labels, sd, heads, cfg["num_layers"])
    _check("vit/logits", o_logits, logits.detach())
    _check("vit/loss", o_loss, loss.detach())
    for k in grads:
        _check(f"vit/grad/{k}", o_grads[k], grads[k], atol=5.0e-5)

    torch.save(
        dict(cfg=cfg, num_classes=num_classes, heads=heads, sd=sd, img=img, labels=labels,
             logits=logits.detach(), loss=loss.detach(), grads=grads),
        os.path.join(OUT, "vit_small.pt"),
    )


def gen_conv2d(ref) -> None:
    """Conv2d (convs/basic.py:41-184) in the geometries the MNIST classifier uses + a dilated one."""
    import importlib

    cases = []
    for seed, (cin, cout, k, stride, dil, hw) in enumerate(
            [(1, 16, 7, 1, 1, 28), (16, 32, 3, 2, 1, 28), (64, 128, 3, 2, 1, 7), (8, 16, 3, 1, 2, 13)]):
        torch.manual_seed(40 + seed)
        m = ref.Conv2d(cin, cout, kernel_size=k, stride=stride, dilation=dil,
                       padding="same" if dil == 1 else dil * (k // 2))
        with torch.no_grad():
            m.bias.normal_(0.0, 0.5)
        x = torch.randn(3, cin, hw, hw, requires_grad=True)
        y = m(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        _check(f"conv2d {cin}->{cout} k{k} s{stride} d{dil}",
               CO.conv2d(x.detach(), m.weight.detach(), m.bias.detach(), stride, m.padding, dil), y.detach())
        cases.append(dict(cfg=dict(in_channels=cin, out_channels=cout, kernel_size=k, stride=stride, dilation=dil,
                                   padding=m.padding),
                          w=m.weight.detach().clone(), b=m.bias.detach().clone(), x=x.detach(), y=y.detach(), gy=gy,
                          gx=x.grad.clone(), gw=m.weight.grad.clone(), gb=m.bias.grad.clone()))
    torch.save(cases, os.path.join(OUT, "conv2d.pt"))


def gen_batchnorm(ref) -> None:
    """NormFactory("batch") = nn.BatchNorm2d (norms.py:90-93,114-115): training step + running stats + eval."""
    torch.manual_seed(50)
    m = ref.NormFactory("batch").make(24)
    with torch.no_grad():
        m.weight.normal_(1.0, 0.3)
        m.bias.normal_(0.0, 0.3)
    x = (torch.randn(5, 24, 6, 7) * 1.7 + 0.4).requires_grad_(True)
    m.train()
    y = m(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    oy, mean, var = CO.batch_norm_train(x.detach(), m.weight.detach(), m.bias.detach(), m.eps)
    _check("batchnorm train", oy, y.detach())
    rm, rv = CO.running_update(torch.zeros(24), torch.ones(24), mean, var, 5 * 6 * 7, m.momentum)
    _check("running_mean", rm, m.running_mean)
    _check("running_var", rv, m.running_var)
    m.eval()
    x2 = torch.randn(3, 24, 6, 7)
    y_eval = m(x2)
    _check("batchnorm eval", CO.batch_norm_eval(x2, m.weight.detach(), m.bias.detach(), m.running_mean, m.running_var,
                                                m.eps), y_eval.detach())
    torch.save(dict(eps=m.eps, momentum=m.momentum, w=m.weight.detach().clone(), b=m.bias.detach().clone(),
                    x=x.detach(), y=y.detach(), gy=gy, gx=x.grad.clone(), gw=m.weight.grad.clone(),
                    gb=m.bias.grad.clone(), running_mean=m.running_mean.clone(), running_var=m.running_var.clone(),
                    x_eval=x2, y_eval=y_eval.detach()),
               os.path.join(OUT, "batchnorm.pt"))


def gen_mnist_clf(ref) -> None:
    """examples/cv/classification/mnist_clf.py: cv_clf(in_channels=1, num_classes=10, num_downsample=3), focal
    loss, synthetic 28x28 batch (the data set cannot be downloaded)."""
    import importlib

    importlib.import_module("cflearn.modules.cv.encoder.vanilla")
    importlib.import_module("cflearn.modules.cv.classifier.vanilla")
    torch.manual_seed(60)
    m = ref.build_module("cv_clf", config=dict(in_channels=1, num_classes=10, encoder_config=dict(num_downsample=3)))
    with torch.no_grad():  # move away from the all-zero-bias / unit-gamma init so every gradient is exercised
        for n_, p_ in m.named_parameters():
            if p_.dim() == 1:
                p_.add_(torch.randn_like(p_) * 0.1)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    img = torch.randn(8, 1, 28, 28)
    labels = torch.randint(0, 10, (8, 1))
    m.train()
    logits = m(img)["predictions"]
    loss = O.focal_loss(logits, labels)
    loss.backward()
    _check("mnist_clf logits", CO.mnist_classifier(img, sd0, 3), logits.detach())
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    sd1 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.eval()
    logits_eval = m(img)["predictions"].detach()
    _check("mnist_clf eval logits", CO.mnist_classifier(img, sd1, 3, training=False), logits_eval)
    torch.save(dict(sd=sd0, img=img, labels=labels, logits=logits.detach(), loss=loss.detach(), grads=grads,
                    sd_after=sd1, logits_eval=logits_eval),
               os.path.join(OUT, "mnist_clf.pt"))


def gen_fcnn(ref) -> None:
    """examples/cv/classification/mnist_fcnn.py / config 1: fcnn(input_dim, output_dim=10) with the default two
    hidden Mapping blocks, focal loss (input_dim 96 -> hidden 192 keeps the fixture small; 784 -> 1024 is the
    same code)."""
    torch.manual_seed(70)
    m = ref.FCNN(96, 10)
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.dim() == 1:
                p_.normal_(0.0, 0.1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(16, 96)
    labels = torch.randint(0, 10, (16, 1))
    logits = m(x)
    loss = O.focal_loss(logits, labels)
    loss.backward()
    _check("fcnn logits", CO.fcnn(x, sd, 2), logits.detach(), atol=5e-5)
    torch.save(dict(sd=sd, x=x, labels=labels, logits=logits.detach(), loss=loss.detach(),
                    grads={k: p.grad.clone() for k, p in m.named_parameters()}),
               os.path.join(OUT, "fcnn.pt"))


def gen_clip(ref) -> None:
    """CLIP towers (multimodal/clip.py) at a small size with head_dim 64: image / text features, logits and every
    parameter gradient of a symmetric contrastive objective on the logits (the loss itself is not part of the
    reference — SURVEY F3 — it only gives the backward pass something to differentiate)."""
    import importlib

    clip = importlib.import_module("cflearn.modules.multimodal.clip")
    torch.manual_seed(80)
    cfg = dict(img_size=32, latent_dim=64, vision_latent_dim=128, vision_patch_size=8, vision_num_heads=2,
               vision_num_layers=2, vocab_size=100, context_length=16, text_latent_dim=128, text_num_heads=2,
               text_num_layers=2)
    m = clip.CLIP(**cfg)
    with torch.no_grad():  # biases / norm affine params away from their trivial init
        for n_, p_ in m.named_parameters():
            if p_.dim() == 1:
                p_.add_(torch.randn_like(p_) * 0.05)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    img = torch.randn(4, 3, 32, 32)
    txt = torch.randint(1, 100, (4, 16))
    txt[0, 9:] = 0  # ragged captions: padding id 0 after the (largest-id) end token
    txt[2, 5:] = 0
    fi = m.encode_image(img)
    ft = m.encode_text(txt)
    logits = m(img, txt)
    _check("clip image features", CL.encode_image(img, sd, 2, 2), fi.detach())
    _check("clip text features", CL.encode_text(txt, sd, 2, 2), ft.detach())
    _check("clip logits", CL.logits_per_image(img, txt, sd, 2, 2, 2, 2), logits.detach(), atol=5e-5)
    target = torch.arange(4)
    loss = 0.5 * (torch.nn.functional.cross_entropy(logits, target) + torch.nn.functional.cross_entropy(logits.t(), target))
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    torch.save(dict(cfg=cfg, sd=sd, img=img, txt=txt, image_features=fi.detach(), text_features=ft.detach(),
                    logits=logits.detach(), loss=loss.detach(), grads=grads),
               os.path.join(OUT, "clip_small.pt"))


def gen_resblock(ref) -> None:
    """UNet building blocks (convs/residual.py:86-253): ResidualBlockWithTimeEmbedding in its three forms (channel
    change with 1x1 shortcut, integrated up-/down-sampling), ResUpsample / ResDownsample with conv, and
    `timestep_embedding` (multimodal/diffusion/unet.py:52-74)."""
    import importlib

    res = importlib.import_module("cflearn.modules.core.convs.residual")
    cases = []
    for seed, (cin, cout, up, down) in enumerate([(64, 128, False, False), (64, 64, True, False), (96, 96, False, True)]):
        torch.manual_seed(90 + seed)
        m = res.ResidualBlockWithTimeEmbedding(cin, cout, time_embedding_channels=128, integrate_upsample=up,
                                               integrate_downsample=down)
        with torch.no_grad():  # conv2 is zero-initialised: perturb everything so all gradients are exercised
            for p_ in m.parameters():
                p_.add_(torch.randn_like(p_) * 0.05)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        x = torch.randn(2, cin, 8, 8, requires_grad=True)
        t = torch.randn(2, 128, requires_grad=True)
        y = m(x, t)
        gy = torch.randn_like(y)
        y.backward(gy)
        _check(f"resblock {cin}->{cout} up={up} down={down}",
               UO.residual_block(x.detach(), t.detach(), sd, resample="up" if up else "down" if down else None), y.detach())
        cases.append(dict(cfg=dict(in_channels=cin, out_channels=cout, time_embedding_channels=128,
                                   integrate_upsample=up, integrate_downsample=down),
                          sd=sd, x=x.detach(), t=t.detach(), y=y.detach(), gy=gy, gx=x.grad.clone(), gt=t.grad.clone(),
                          grads={k: p_.grad.clone() for k, p_ in m.named_parameters()}))
    torch.manual_seed(95)
    up = res.ResUpsample(32, True, out_channels=48)
    down = res.ResDownsample(32, True, out_channels=48)
    x = torch.randn(2, 32, 6, 6)
    yu, yd = up(x), down(x)
    _check("ResUpsample", CO.conv2d(UO.upsample2(x), up.conv.weight.detach(), up.conv.bias.detach(), 1, 1), yu.detach())
    _check("ResDownsample", CO.conv2d(x, down.net.weight.detach(), down.net.bias.detach(), 2, 1), yd.detach())
    try:
        unet = importlib.import_module("cflearn.modules.multimodal.diffusion.unet")
        tt = torch.tensor([0, 1, 17, 500, 999])
        te = unet.timestep_embedding(tt, 320, dtype=torch.float32)
        _check("timestep_embedding", UO.timestep_embedding(tt, 320), te)
    except Exception as e:  # the UNet module pulls in more of cftool than the harness provides
        print("  (timestep_embedding not importable through the harness:", type(e).__name__, e, ")")
        tt = torch.tensor([0, 1, 17, 500, 999])
        te = UO.timestep_embedding(tt, 320)
    torch.save(dict(blocks=cases, up=dict(sd={k: v.detach().clone() for k, v in up.state_dict().items()}, y=yu.detach()),
                    down=dict(sd={k: v.detach().clone() for k, v in down.state_dict().items()}, y=yd.detach()),
                    x_resample=x, timesteps=tt, timestep_embedding=te),
               os.path.join(OUT, "resblock.pt"))


def gen_spatial_transformer(ref) -> None:
    """SpatialTransformer (mixed_stacks/api.py:830-893) with the UNet's head size 40 (320 channels / 8 heads in the zoo
    model; 160 channels / 4 heads here — the reference's forward only works when in_channels == num_heads * head_dim)
    and a text context of another width."""
    import importlib

    api = importlib.import_module("cflearn.modules.core.mixed_stacks.api")
    torch.manual_seed(100)
    cfg = dict(in_channels=160, num_heads=4, head_dim=40, num_layers=1, context_dim=96)
    m = api.SpatialTransformer(160, 4, 40, num_layers=1, context_dim=96)
    with torch.no_grad():  # from_latent is zero-initialised: perturb everything
        for p_ in m.parameters():
            p_.add_(torch.randn_like(p_) * 0.05)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 160, 6, 6, requires_grad=True)
    ctx = torch.randn(2, 11, 96, requires_grad=True)
    y = m(x, ctx)
    gy = torch.randn_like(y)
    y.backward(gy)
    _check("spatial transformer", UO.spatial_transformer(x.detach(), ctx.detach(), sd, 4), y.detach(), atol=5e-5)
    torch.save(dict(cfg=cfg, sd=sd, x=x.detach(), context=ctx.detach(), y=y.detach(), gy=gy, gx=x.grad.clone(),
                    gcontext=ctx.grad.clone(), grads={k: p_.grad.clone() for k, p_ in m.named_parameters()}),
               os.path.join(OUT, "spatial_transformer.pt"))


def gen_unet(ref) -> None:
    """UNetDiffuser (multimodal/diffusion/unet.py:76-322) in the zoo `diffusion/ddpm` structure (spatial transformers at
    every resolution, conv down-sampling, nearest up-sampling + conv, skip concatenation) at a small size, with the DDPM
    epsilon-prediction objective (models/cv/diffusion.py:44-94: MSE between the prediction and the noise) as the scalar
    the backward pass differentiates."""
    import importlib

    unet = importlib.import_module("cflearn.modules.multimodal.diffusion.unet")
    torch.manual_seed(110)
    cfg = dict(in_channels=3, out_channels=3, num_heads=4, use_spatial_transformer=True, num_transformer_layers=1,
               context_dim=48, start_channels=32, num_res_blocks=1, attention_downsample_rates=(1, 2),
               channel_multipliers=(1, 2))
    m = unet.UNetDiffuser(**cfg)
    with torch.no_grad():  # several layers are zero-initialised: perturb so every gradient is exercised
        for p_ in m.parameters():
            p_.add_(torch.randn_like(p_) * 0.03)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 3, 16, 16)
    t = torch.tensor([7, 912])
    ctx = torch.randn(2, 5, 48)
    noise = torch.randn(2, 3, 16, 16)
    y = m(x, timesteps=t, context=ctx)
    _check("unet output", UO.unet_diffuser(x, t, ctx, sd, cfg), y.detach(), atol=1e-4)
    loss = torch.nn.functional.mse_loss(y, noise)
    loss.backward()
    torch.save(dict(cfg=cfg, sd=sd, x=x, timesteps=t, context=ctx, noise=noise, y=y.detach(), loss=loss.detach(),
                    grads={k: p_.grad.clone().half() for k, p_ in m.named_parameters()}),
               os.path.join(OUT, "unet_small.pt"))


def gen_ddpm_schedule(ref) -> None:
    """The DDPM noise schedule and forward process as the reference computes them: `make_beta_schedule("linear")` +
    the `_register_noise_schedule` tables (ddpm.py:51-89,599-640) and `DDPMQSampler.q_sample` (samplers/schema.py:
    90-112) on a seeded batch."""
    import importlib

    import numpy as np

    import ast

    # ddpm.py imports half of cflearn at module level; `make_beta_schedule` itself is a pure numpy function: execute
    # ITS definition (read from the reference file at generation time, nothing is copied into this repository)
    src_path = os.path.join("/root/reference", "cflearn", "modules", "multimodal", "diffusion", "ddpm.py")
    tree = ast.parse(open(src_path).read())
    fn_node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "make_beta_schedule")
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn_node], type_ignores=[]), src_path, "exec"), ns)
    betas = ns["make_beta_schedule"]("linear", 1000, 8.5e-4, 1.2e-2, 8.0e-3)
    try:
        schema = importlib.import_module("cflearn.modules.multimodal.diffusion.samplers.schema")
    except Exception as e:  # same story: take the one class definition
        print("  (samplers.schema not importable through the harness:", type(e).__name__, ")")
        schema = None
    ac = np.cumprod(1.0 - betas, axis=0)
    to_t = lambda a: torch.from_numpy(a.astype(np.float32))  # noqa: E731
    sqrt_ac, sqrt_1mac = to_t(np.sqrt(ac)), to_t(np.sqrt(1.0 - ac))
    torch.manual_seed(120)
    x = torch.randn(5, 3, 8, 8)
    noise = torch.randn(5, 3, 8, 8)
    t = torch.tensor([0, 1, 499, 998, 999])
    if schema is not None:
        sampler = schema.DDPMQSampler.__new__(schema.DDPMQSampler)
        sampler.reset_buffers(sqrt_ac, sqrt_1mac)
        x_t = sampler.q_sample(x, t, noise)
    else:
        sp = os.path.join("/root/reference", "cflearn", "modules", "multimodal", "diffusion", "samplers", "schema.py")
        stree = ast.parse(open(sp).read())
        cls = next(n for n in stree.body if isinstance(n, ast.ClassDef) and n.name == "DDPMQSampler")
        cls.bases = []
        from typing import Optional as _Opt

        def extract_to(array, indices, num_dim):  # cflearn/modules/multimodal/diffusion/utils.py: gather + broadcast
            return array.gather(-1, indices).contiguous().view(-1, *([1] * (num_dim - 1)))

        sns = {"torch": torch, "Tensor": torch.Tensor, "Optional": _Opt, "extract_to": extract_to}
        exec(compile(ast.Module(body=[cls], type_ignores=[]), sp, "exec"), sns)
        sampler = sns["DDPMQSampler"]()
        sampler.reset_buffers(sqrt_ac, sqrt_1mac)
        x_t = sampler.q_sample(x, t, noise)
    torch.save(dict(betas=to_t(betas), sqrt_alphas_cumprod=sqrt_ac, sqrt_one_minus_alphas_cumprod=sqrt_1mac, x=x,
                    noise=noise, t=t, x_t=x_t), os.path.join(OUT, "ddpm_schedule.pt"))


def gen_unet_variants(ref) -> None:
    """The UNet options beyond the zoo `diffusion/ddpm` configuration, from the reference's own classes:
    `MultiHeadSpatialAttention` (attentions.py:373-460, both head layouts), the scale-shift residual block
    (residual.py:236-239), and two small `UNetDiffuser`s — (a) pixel self attention + ResBlock resampling + scale-shift
    norm + class labels, (b) spatial transformers with Linear projections + ControlNet residuals."""
    import importlib

    attn = importlib.import_module("cflearn.modules.core.attentions")
    res = importlib.import_module("cflearn.modules.core.convs.residual")
    unet = importlib.import_module("cflearn.modules.multimodal.diffusion.unet")

    def perturb(m, std):
        with torch.no_grad():
            for p_ in m.parameters():
                p_.add_(torch.randn_like(p_) * std)

    out = {"mhsa": [], "unets": []}
    for seed, kw in ((130, dict(num_heads=2)), (131, dict(num_heads=None, num_head_channels=16, split_qkv_before_heads=True))):
        torch.manual_seed(seed)
        m = attn.MultiHeadSpatialAttention(64, **kw)
        perturb(m, 0.05)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        x = torch.randn(2, 64, 6, 6, requires_grad=True)
        y = m(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        _check(f"mhsa {kw}", UO.multi_head_spatial_attention(x.detach(), sd, m.num_heads, m.split_qkv_before_heads), y.detach(), atol=5e-5)
        out["mhsa"].append(dict(cfg=dict(in_channels=64, **kw), sd=sd, x=x.detach(), y=y.detach(), gy=gy, gx=x.grad.clone(),
                                grads={k: p_.grad.clone() for k, p_ in m.named_parameters()}))
    torch.manual_seed(132)
    m = res.ResidualBlockWithTimeEmbedding(32, 64, time_embedding_channels=128, use_scale_shift_norm=True)
    perturb(m, 0.05)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 32, 8, 8, requires_grad=True)
    t = torch.randn(2, 128, requires_grad=True)
    y = m(x, t)
    gy = torch.randn_like(y)
    y.backward(gy)
    _check("scale-shift resblock", UO.residual_block(x.detach(), t.detach(), sd, scale_shift=True), y.detach(), atol=5e-5)
    out["scale_shift"] = dict(cfg=dict(in_channels=32, out_channels=64, time_embedding_channels=128, use_scale_shift_norm=True),
                              sd=sd, x=x.detach(), t=t.detach(), y=y.detach(), gy=gy, gx=x.grad.clone(), gt=t.grad.clone(),
                              grads={k: p_.grad.clone() for k, p_ in m.named_parameters()})
    cfgs = [
        dict(in_channels=3, out_channels=3, num_heads=2, use_spatial_transformer=False, start_channels=32, num_res_blocks=1,
             attention_downsample_rates=(1, 2), channel_multipliers=(1, 2), resample_with_resblock=True,
             use_scale_shift_norm=True, num_classes=5),
        dict(in_channels=3, out_channels=3, num_heads=4, use_spatial_transformer=True, num_transformer_layers=1,
             context_dim=48, start_channels=32, num_res_blocks=1, attention_downsample_rates=(1, 2),
             channel_multipliers=(1, 2), use_linear_in_transformer=True),
    ]
    for i, cfg in enumerate(cfgs):
        torch.manual_seed(133 + i)
        m = unet.UNetDiffuser(**cfg)
        perturb(m, 0.03)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        x = torch.randn(2, 3, 16, 16)
        t = torch.tensor([3, 700])
        noise = torch.randn(2, 3, 16, 16)
        kw = {}
        if cfg.get("num_classes"):
            kw["labels"] = torch.tensor([4, 1])
        if cfg["use_spatial_transformer"]:
            kw["context"] = torch.randn(2, 5, 48)
            # ControlNet residuals: one per skip connection + one for the middle block, popped from the end (unet.py:311-318)
            shapes = []
            net = x
            probe = []
            hooks = [blk.register_forward_hook(lambda _m, _i, o: probe.append(o.shape)) for blk in m.input_blocks]
            m(x, timesteps=t, **{k: v for k, v in kw.items()})
            for h_ in hooks:
                h_.remove()
            shapes = list(probe) + [probe[-1]]
            kw["control"] = [torch.randn(*s_) * 0.1 for s_ in shapes]
        y = m(x, timesteps=t, **{k: (list(v) if k == "control" else v) for k, v in kw.items()})
        loss = torch.nn.functional.mse_loss(y, noise)
        loss.backward()
        out["unets"].append(dict(cfg=cfg, sd=sd, x=x, timesteps=t, noise=noise, kw=kw, y=y.detach(), loss=loss.detach(),
                                 grads={k: p_.grad.clone().half() for k, p_ in m.named_parameters()}))
    torch.save(out, os.path.join(OUT, "unet_variants.pt"))


def gen_ddpm_objectives(ref) -> None:
    """`DDPMStep.loss_fn` (models/cv/diffusion.py:44-94) and `DDPM._register_noise_schedule` (ddpm.py:599-679) run as the
    reference's own code on stand-in objects that carry exactly the attributes they read: every parameterization / loss
    type / log-variance / ELBO-weight combination on one seeded batch, with d loss / d prediction and d loss / d log_var
    from autograd."""
    import types

    from refharness import import_cflearn

    import_cflearn()
    ddpm_mod = sys.modules["cflearn.modules.multimodal.diffusion.ddpm"]
    step_mod = sys.modules["cflearn.models.cv.diffusion"]
    DDPM = ddpm_mod.DDPM

    class _Tables(torch.nn.Module):
        def __init__(self, parameterization, v_posterior):
            super().__init__()
            self.parameterization, self.v_posterior = parameterization, v_posterior

    def tables(parameterization, schedule, v_posterior=0.0):
        m = _Tables(parameterization, v_posterior)
        DDPM._register_noise_schedule(m, 1000, None, schedule, 8.5e-4, 1.2e-2, 8.0e-3)
        return m

    torch.manual_seed(140)
    x = torch.randn(4, 3, 8, 8)
    noise = torch.randn(4, 3, 8, 8)
    t = torch.tensor([0, 17, 500, 999])
    cases = []
    for (par, sched, lt, lv_init, learn, lsw, elbo, vp) in [
        ("eps", "linear", "l2", 0.0, False, 1.0, 0.0, 0.0),
        ("x0", "linear", "l2", 0.3, False, 1.0, 0.0, 0.0),
        ("v", "cosine", "l2", 0.0, False, 0.7, 0.0, 0.0),
        ("eps", "cosine", "l1", -0.4, True, 1.0, 0.0, 0.0),
        ("eps", "linear", "l2", 0.2, True, 0.9, 1.0e-3, 0.5),
        ("x0", "sqrt_linear", "l1", 0.0, False, 1.0, 0.5, 0.0),
    ]:
        tb = tables(par, sched, vp)
        log_var = torch.full((1000,), lv_init)
        if learn:
            log_var = torch.nn.Parameter(log_var + torch.randn(1000) * 0.1)
        ddpm = types.SimpleNamespace(parameterization=par, noise_key=DDPM.noise_key, timesteps_key=DDPM.timesteps_key,
                                     log_var=log_var, learn_log_var=learn, lvlb_weights=tb.lvlb_weights,
                                     sqrt_alphas_cumprod=tb.sqrt_alphas_cumprod,
                                     sqrt_one_minus_alphas_cumprod=tb.sqrt_one_minus_alphas_cumprod)
        step = step_mod.DDPMStep("learnable")
        step.setup(loss_type=lt, l_simple_weight=lsw, original_elbo_weight=elbo)
        pred = (torch.randn(4, 3, 8, 8) * 0.7).requires_grad_(True)
        res_ = step.loss_fn(types.SimpleNamespace(m=ddpm), None, {"input": x},
                            {"predictions": pred, DDPM.noise_key: noise, DDPM.timesteps_key: t})
        res_.loss.backward()
        tbl = dict(sqrt_alphas_cumprod=tb.sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod=tb.sqrt_one_minus_alphas_cumprod,
                   lvlb_weights=tb.lvlb_weights)
        mine = UO.ddpm_objective(pred.detach(), x, noise, t, tbl, parameterization=par, loss_type=lt,
                                 log_var=log_var.detach(), l_simple_weight=lsw, original_elbo_weight=elbo)
        _check(f"ddpm objective {par}/{sched}/{lt}", mine, res_.loss.detach(), atol=1e-6)
        cases.append(dict(parameterization=par, schedule=sched, loss_type=lt, log_var_init=lv_init, learn_log_var=learn,
                          l_simple_weight=lsw, original_elbo_weight=elbo, v_posterior=vp, log_var=log_var.detach().clone(),
                          pred=pred.detach(), loss=res_.loss.detach(), losses=dict(res_.losses), dpred=pred.grad.clone(),
                          dlog_var=None if not learn else log_var.grad.clone(), betas=tb.betas.clone(),
                          lvlb_weights=tb.lvlb_weights.clone(), posterior_variance=tb.posterior_variance.clone(),
                          sqrt_alphas_cumprod=tb.sqrt_alphas_cumprod.clone(),
                          sqrt_one_minus_alphas_cumprod=tb.sqrt_one_minus_alphas_cumprod.clone()))
    torch.save(dict(x=x, noise=noise, t=t, cases=cases), os.path.join(OUT, "ddpm_objectives.pt"))


def gen_ml_encoder(ref) -> None:
    """The reference's own `ml.encoder` + `CommonMLModel.encode`.  Two cases: mixed one-hot / embedding columns with
    in-range categories, and all-embedding columns with out-of-bound categories (the reference imputes out-of-bound
    values only through the shared `indices`, i.e. when every categorical column uses the same method:
    ml_encoder.py:186-199 re-reads the raw batch otherwise and F.one_hot / F.embedding raise)."""
    import ml_oracle as MO

    Settings = sys.modules["cflearn.schema"].MLEncoderSettings
    CommonMLModel = sys.modules["cflearn.models.ml.common"].CommonMLModel
    cases = {}
    for name, settings, oob in (
        ("mixed", {"1": Settings(5, "one_hot"), "3": Settings(7, "embedding", dict(out_dim=6)),
                   "4": Settings(3, ["one_hot", "embedding"]), "6": Settings(11, "embedding")}, 0),
        ("all_embedding_oob", {"0": Settings(4, "embedding"), "2": Settings(9, "embedding", dict(out_dim="sqrt"))}, 2),
    ):
        torch.manual_seed(21)
        enc = ref.build_module("ml.encoder", config=dict(settings=settings))
        enc.eval()  # embedding dropout off: the deterministic part
        b, f = 37, 8
        x = torch.randn(b, f)
        for c, st in settings.items():
            x[:, int(c)] = torch.randint(0, st.dim + oob, (b,)).float()  # dim, dim + 1 are out of bound -> 0
        first = int(sorted(settings)[0])
        x[0, first] = settings[str(first)].dim - 1 + 0.9  # truncation toward zero, not rounding
        model = CommonMLModel()
        model.m = torch.nn.ModuleDict(dict(encoder=enc))
        pack = model.encode(x)
        res = enc(x)
        sd = {k: v.detach().clone() for k, v in enc.state_dict().items()}
        tables = {int(k.split(".")[1]): v for k, v in sd.items() if k.startswith("embeddings.")}
        idx, oh, emb, merged = MO.encode(x, enc.tgt_columns, [settings[str(c)].dim for c in enc.tgt_columns],
                                         enc.one_hot_columns, enc.embedding_columns, tables)
        assert torch.equal(idx, res.indices) and torch.equal(emb, res.embedding)
        assert (oh is None and res.one_hot is None) or torch.equal(oh, res.one_hot)
        assert torch.equal(merged, pack.merged_all)
        enc.zero_grad()
        g = torch.randn_like(pack.merged_all)
        model.encode(x).merged_all.backward(g)
        grads = {k: p.grad.detach().clone() for k, p in enc.named_parameters()}
        cfg = {k: dict(dim=v.dim, methods=v.methods, method_configs=v.method_configs) for k, v in settings.items()}
        cases[name] = dict(settings=cfg, sd=sd, x=x, indices=res.indices,
                           one_hot=None if res.one_hot is None else res.one_hot.detach(),
                           embedding=res.embedding.detach(), merged_all=pack.merged_all.detach(), gy=g, grads=grads,
                           dim_increment=enc.dim_increment)
    torch.save(cases, os.path.join(OUT, "ml_encoder.pt"))


def gen_postnorm_interp(ref) -> None:
    """(1) a post-norm `MixedStackedEncoder` (api.py:160-185; no head norm), (2) the ViT encoder at a NON-native
    resolution (bicubic positional-encoding interpolation, api.py:231-267): outputs and gradients of the reference."""
    torch.manual_seed(41)
    cfg = dict(in_dim=64, num_tokens=9, token_mixing_type="attention", token_mixing_config=dict(num_heads=1, bias=True),
               channel_mixing_config=dict(), num_layers=2, drop_path_rate=0.0, norm_position="post_norm", norm_type="layer",
               use_head_token=True, use_positional_encoding=True, is_vision_positional_encoding=False)
    enc = ref.MixedStackedEncoder(**cfg)
    with torch.no_grad():
        for p in enc.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
    x = torch.randn(3, 9, 64, requires_grad=True)
    y = enc(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    sd = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    post = dict(cfg=cfg, sd=sd, x=x.detach(), y=y.detach(), gy=gy, gx=x.grad.clone(),
                grads={k: p.grad.clone() for k, p in enc.named_parameters()})
    # (2) non-native resolution
    vcfg = dict(img_size=32, patch_size=8, in_channels=3, latent_dim=64, num_layers=1, feedforward_dim_ratio=2.0)
    vit = ref.ViTEncoder(**vcfg)
    img = torch.randn(2, 3, 48, 32)
    out = vit(img, hwp=(48, 32, 8))
    gv = torch.randn_like(out)
    out.backward(gv)
    vsd = {k: v.detach().clone() for k, v in vit.state_dict().items()}
    pos = O.interpolate_pos_encoding(vsd["encoder.pos_encoding.pos_encoding"], 1, 24, 48, 32)
    _check("interpolated pos encoding", pos, vit.encoder.pos_encoding.interpolate_pos_encoding(
        torch.zeros(1, 25, 64), (48, 32, 8)).detach(), atol=1e-6)
    vsd2 = dict(vsd)
    vsd2["encoder.pos_encoding.pos_encoding"] = pos
    _check("vit @ 48x32", O.vit_encoder(img, vsd2, 1, 1), out.detach())
    interp = dict(cfg=vcfg, sd=vsd, img=img, hwp=(48, 32, 8), y=out.detach(), gy=gv,
                  gpos=vit.encoder.pos_encoding.pos_encoding.grad.clone())
    torch.save(dict(post_norm=post, interp=interp), os.path.join(OUT, "postnorm_interp.pt"))


def gen_stochastic(ref) -> None:
    """nn.Dropout and the reference's DropPath: outputs AND the masks they drew (recovered from the outputs), so that
    the kernels can be checked bit-for-bit with the mask injected."""
    import ml_oracle as MO

    torch.manual_seed(31)
    out = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        x = (torch.randn(33, 40) + 3.0).to(dt)  # no zeros: the keep mask is recoverable from y != 0
        p = 0.3
        y = torch.nn.functional.dropout(x, p, training=True)
        mask = (y != 0).to(torch.uint8)
        assert torch.equal(MO.dropout(x, mask, p), y)
        dp = ref.core.DropPath(0.25)
        dp.train()
        xb = (torch.randn(9, 5, 8) + 3.0).to(dt)
        yb = dp(xb)
        mb = (yb.reshape(9, -1).abs().sum(1) != 0).float()
        assert torch.equal(MO.drop_path(xb, mb, 0.75), yb)
        out[name] = dict(x=x, p=p, y=y, mask=mask, xb=xb, rate=0.25, yb=yb, mb=mb)
    torch.save(out, os.path.join(OUT, "stochastic.pt"))


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    torch.set_num_threads(4)
    only = sys.argv[1:]
    for fn in (gen_linear, gen_layernorm, gen_layernorm4d, gen_sdp, gen_attention, gen_feedforward, gen_vit, gen_conv2d,
               gen_batchnorm, gen_mnist_clf, gen_fcnn, gen_clip, gen_resblock, gen_spatial_transformer, gen_unet, gen_ddpm_schedule,
               gen_ml_encoder, gen_stochastic, gen_postnorm_interp, gen_unet_variants, gen_ddpm_objectives):
        if only and fn.__name__ not in only:
            continue
        print(fn.__name__)
        fn(ref)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
