"""Conv2d (im2row + MFMA GEMM) / BatchNorm / LeakyReLU / pooling / focal loss and the two models built from
them (MNIST conv classifier, FCNN) on the GPU vs the reference-made golden fixtures and the CPU oracle.

Tolerances: the HIP path rounds activations to bf16 between layers (what the reference does under
mixed_precision="bf16"); fixtures are the reference's fp32 CPU run.  Index work (im2row / row2im / transposes) is
checked bit-exactly on bf16-representable data."""
import pytest
import torch

import cflearn_amd as C
from cflearn_amd import functional as HF
from cflearn_amd import ops
from helpers import assert_close, bf16_round

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def test_im2row_row2im_transposes_bit_exact():
    import conv_oracle as CO

    torch.manual_seed(0)
    for (b, c, h, w, k, s, p, d) in ((2, 3, 9, 11, 3, 1, 1, 1), (3, 1, 28, 28, 7, 1, 3, 1), (2, 16, 14, 14, 3, 2, 1, 1),
                                    (1, 5, 13, 10, 3, 2, 2, 2), (2, 8, 7, 7, 1, 1, 0, 1)):
        x = bf16_round(torch.randn(b, c, h, w))
        ho, wo = ops.conv_out_hw(h, w, k, k, s, p, d)
        rows = ops.conv_im2row(x.to(DEV), k, k, s, p, d)
        kk = c * k * k
        # oracle rows through an identity "weight": conv with one-hot kernels picks the patch elements
        eye = torch.eye(kk).view(kk, c, k, k)
        want = CO.conv2d(x, eye, None, s, p, d).permute(0, 2, 3, 1).reshape(b * ho * wo, kk)
        assert torch.equal(rows[:, :kk].float().cpu(), want)
        assert rows.shape[1] % 8 == 0 and (rows[:, kk:] == 0).all()
        # same from a bf16 input
        assert torch.equal(ops.conv_im2row(x.to(DEV).bfloat16(), k, k, s, p, d), rows)
        # row2im is the transpose of im2row: <im2row(x), r> == <x, row2im(r)> — check against autograd of the oracle
        r = bf16_round(torch.randn(b * ho * wo, rows.shape[1]) * 0.25)
        r[:, kk:] = 0
        xg = x.clone().requires_grad_(True)
        CO.conv2d(xg, eye, None, s, p, d).permute(0, 2, 3, 1).reshape(b * ho * wo, kk).backward(r[:, :kk])
        got = ops.conv_row2im(r.to(DEV).bfloat16(), (b, c, h, w), k, k, s, p, d)
        assert_close(got, xg.grad, 4e-3, "row2im")  # sums of <= k*k bf16 values, rounded once to bf16
    t = bf16_round(torch.randn(5, 37, 70))
    assert torch.equal(ops.transpose_batched(t.to(DEV)).float().cpu(), t.transpose(1, 2))
    assert torch.equal(ops.transpose_batched(t.to(DEV).bfloat16()).float().cpu(), t.transpose(1, 2))
    # the 16-bytes-per-lane form (R % 8 == 0 and C % 8 == 0): partial edge tiles, f32 and bf16 sources, a batch
    for n, r, c in ((3, 72, 200), (2, 4096, 320), (1, 8, 8), (5, 64, 1280), (2, 136, 64)):
        t = bf16_round(torch.randn(n, r, c))
        assert torch.equal(ops.transpose_batched(t.to(DEV)).float().cpu(), t.transpose(1, 2)), (n, r, c)
        assert torch.equal(ops.transpose_batched(t.to(DEV).bfloat16()).float().cpu(), t.transpose(1, 2)), (n, r, c)


def test_conv2d_golden(golden):
    for case in golden("conv2d.pt"):
        cfg = case["cfg"]
        m = C.Conv2d(cfg["in_channels"], cfg["out_channels"], kernel_size=cfg["kernel_size"], stride=cfg["stride"],
                     dilation=cfg["dilation"], padding=cfg["padding"]).to(DEV)
        with torch.no_grad():
            m.weight.copy_(case["w"])
            m.bias.copy_(case["b"])
        x = case["x"].to(DEV).requires_grad_(True)
        y = m(x)
        assert y.dtype == torch.bfloat16 and y.shape == case["y"].shape
        assert_close(y, case["y"], 6e-3, f"conv y {cfg}")
        y.backward(case["gy"].to(DEV).bfloat16())
        assert_close(x.grad, case["gx"], 1e-2, f"conv gx {cfg}")
        assert_close(m.weight.grad, case["gw"], 1e-2, f"conv gw {cfg}")
        assert_close(m.bias.grad, case["gb"], 6e-3, f"conv gb {cfg}")
        # accumulate into existing grads
        y2 = m(x.detach())
        y2.backward(case["gy"].to(DEV).bfloat16())
        assert_close(m.weight.grad, 2 * case["gw"], 1e-2, "conv gw accumulate")


def test_batchnorm_golden(golden):
    g = golden("batchnorm.pt")
    m = C.NormFactory("batch").make(g["w"].numel()).to(DEV)
    assert isinstance(m, torch.nn.BatchNorm2d) and m.eps == g["eps"] and m.momentum == g["momentum"]
    with torch.no_grad():
        m.weight.copy_(g["w"])
        m.bias.copy_(g["b"])
    x = g["x"].to(DEV).requires_grad_(True)
    m.train()
    y = m(x)
    assert_close(y, g["y"], 4e-3, "bn y")
    y.backward(g["gy"].to(DEV).bfloat16())
    assert_close(x.grad, g["gx"], 8e-3, "bn gx")
    assert_close(m.weight.grad, g["gw"], 6e-3, "bn gw")
    assert_close(m.bias.grad, g["gb"], 6e-3, "bn gb")
    assert_close(m.running_mean, g["running_mean"], 1e-5, "running_mean")
    assert_close(m.running_var, g["running_var"], 1e-5, "running_var")
    assert int(m.num_batches_tracked) == 1
    m.eval()
    assert_close(m(g["x_eval"].to(DEV)), g["y_eval"], 4e-3, "bn eval")
    # BatchNorm1d forms: [B, C] and the token-major BN wrapper
    torch.manual_seed(3)
    bn1 = C.NormFactory("batch1d").make(40).to(DEV)
    ref1 = torch.nn.BatchNorm1d(40)
    x1 = torch.randn(33, 40) * 2 + 1
    assert_close(bn1(x1.to(DEV)), ref1(x1), 4e-3, "bn1d")
    assert_close(bn1.running_var, ref1.running_var, 1e-5, "bn1d running_var")
    bnt = C.NormFactory("batch_norm").make(24).to(DEV)
    xt = torch.randn(4, 9, 24)
    reft = torch.nn.BatchNorm1d(24)
    assert_close(bnt(xt.to(DEV)), reft(xt.transpose(1, 2)).transpose(1, 2), 4e-3, "BN token-major")


def test_activations_pool_focal():
    import vit_oracle as O

    torch.manual_seed(1)
    x = bf16_round(torch.randn(3, 5, 7, 9))
    for slope, act in ((0.2, C.modules.build_activation("leaky_relu_0.2")), (0.0, C.modules.build_activation("ReLU"))):
        xg = x.to(DEV).bfloat16().requires_grad_(True)
        y = act(xg)
        want = torch.where(x > 0, x, x * slope)
        assert torch.equal(y.float().cpu(), bf16_round(want))
        gy = bf16_round(torch.randn_like(x))
        y.backward(gy.to(DEV).bfloat16())
        assert torch.equal(xg.grad.float().cpu(), bf16_round(torch.where(x > 0, gy, gy * slope)))
    xg = x.to(DEV).bfloat16().requires_grad_(True)
    p = HF.global_avg_pool(xg)
    assert_close(p, x.mean(dim=(2, 3)), 4e-3, "avgpool")
    gy = bf16_round(torch.randn(3, 5))
    p.backward(gy.to(DEV).bfloat16())
    assert_close(xg.grad, (gy / 63.0)[:, :, None, None].expand_as(x), 4e-3, "avgpool bwd")
    # focal loss: value and gradient vs the fp32 oracle (labels: integer gather, exact)
    logits = torch.randn(37, 10) * 2
    labels = torch.randint(0, 10, (37, 1))
    lg = logits.clone().requires_grad_(True)
    want = O.focal_loss(lg, labels)
    want.backward()
    l2 = logits.to(DEV).requires_grad_(True)
    got = HF.focal_loss(l2, labels.to(DEV))
    got.backward()
    assert abs(got.item() - want.item()) <= 1e-5 * max(1.0, abs(want.item()))
    assert_close(l2.grad, lg.grad, 1e-5, "focal dlogits")


def _grads(m):
    return {k: p.grad.detach().float().cpu() for k, p in m.named_parameters() if p.grad is not None}


def test_mnist_conv_classifier_golden(golden):
    """examples/cv/classification/mnist_clf.py on a synthetic 28x28 batch: logits / focal loss / every parameter
    gradient / running statistics after the step / eval-mode logits vs the reference's fp32 CPU run."""
    g = golden("mnist_clf.pt")
    m = C.build_module("cv_clf", config=dict(in_channels=1, num_classes=10, encoder_config=dict(num_downsample=3)))
    m.load_state_dict(g["sd"])
    m = m.to(DEV).train()
    logits = m(g["img"].to(DEV))["predictions"]
    assert logits.dtype == torch.float32
    assert_close(logits, g["logits"], 2e-2, "logits")
    loss = HF.focal_loss(logits, g["labels"].to(DEV))
    assert abs(loss.item() - g["loss"].item()) <= 5e-3 * abs(g["loss"].item())
    loss.backward()
    grads = _grads(m)
    assert set(grads) == set(g["grads"])
    # (1) vs the oracle with bf16 storage emulated at the points where the HIP path stores bf16: kernel error only
    import conv_oracle as CO
    import vit_oracle as O

    params = {k: v.clone().requires_grad_(True) for k, v in g["sd"].items() if k in g["grads"]}
    buffers = {k: v for k, v in g["sd"].items() if k not in params}
    O.focal_loss(CO.mnist_classifier(g["img"], {**buffers, **params}, 3, rnd=CO.bf16_round), g["labels"]).backward()
    worst = 0.0
    for k, v in grads.items():
        worst = max(worst, assert_close(v, params[k].grad, 2.5e-2, f"grad {k} vs bf16-storage oracle", abs_floor=2e-4))
    # (2) vs the reference's fp32 run: bf16 storage moves the deepest gradients by several percent on this model (the
    # emulated oracle above differs from fp32 by the same amount), so this bound is loose by construction
    for k, v in grads.items():
        assert_close(v, g["grads"][k], 1.2e-1, f"grad {k} vs fp32 reference", abs_floor=2e-4)
    print(f"mnist_clf worst grad rel-L2 vs bf16-storage oracle {worst:.3e}")
    sd = m.state_dict()
    for k, v in g["sd_after"].items():
        if "running" in k:
            assert_close(sd[k], v, 1e-2, k, abs_floor=1e-4)
        if "num_batches_tracked" in k:
            assert int(sd[k]) == int(v)
    m.load_state_dict(g["sd_after"])
    m.eval()
    assert_close(m(g["img"].to(DEV))["predictions"], g["logits_eval"], 2e-2, "eval logits")


def test_fcnn_golden(golden):
    g = golden("fcnn.pt")
    m = C.build_module("fcnn", config=dict(input_dim=96, output_dim=10))
    m.load_state_dict(g["sd"])
    m = m.to(DEV)
    logits = m(g["x"].to(DEV))
    assert logits.dtype == torch.float32
    assert_close(logits, g["logits"], 1e-2, "fcnn logits")
    loss = HF.focal_loss(logits, g["labels"].to(DEV))
    assert abs(loss.item() - g["loss"].item()) <= 3e-3 * abs(g["loss"].item())
    loss.backward()
    import conv_oracle as CO
    import vit_oracle as O

    sd = {k: v.clone().requires_grad_(True) for k, v in g["sd"].items()}
    O.focal_loss(CO.fcnn(g["x"], sd, 2, rnd=CO.bf16_round), g["labels"]).backward()
    for k, v in _grads(m).items():
        assert_close(v, sd[k].grad, 1.5e-2, f"fcnn grad {k} vs bf16-storage oracle", abs_floor=1e-4)
        assert_close(v, g["grads"][k], 1.2e-1, f"fcnn grad {k} vs fp32 reference", abs_floor=1e-4)


def test_conv_classifier_and_fcnn_train_steps(golden):
    """engine.TrainStep (fused AdamW over the arena, focal loss kernel) drives both models: the loss of a fixed
    batch goes down, BatchNorm running statistics move, nothing is NaN."""
    from cflearn_amd.engine import TrainStep

    g = golden("mnist_clf.pt")
    m = C.build_module("cv_clf", config=dict(in_channels=1, num_classes=10, encoder_config=dict(num_downsample=3)))
    m.load_state_dict(g["sd"])
    m = m.to(DEV).train()
    ts = TrainStep(m, lr=2e-3, loss="focal")
    img, labels = g["img"].to(DEV), g["labels"].view(-1).to(DEV)
    losses = [ts.step(img, labels).item() / img.shape[0] for _ in range(25)]
    assert abs(losses[0] - g["loss"].item()) <= 5e-3 * g["loss"].item()
    assert losses[-1] < 0.5 * losses[0] and all(l == l for l in losses)
    assert int(m.encoder.encoder.encoder[1].num_batches_tracked) == 25

    f = golden("fcnn.pt")
    n = C.build_module("fcnn", config=dict(input_dim=96, output_dim=10))
    n.load_state_dict(f["sd"])
    n = n.to(DEV)
    ts2 = TrainStep(n, lr=2e-3, loss="focal")
    x, y = f["x"].to(DEV), f["labels"].view(-1).to(DEV)
    losses = [ts2.step(x, y).item() / x.shape[0] for _ in range(25)]
    assert abs(losses[0] - f["loss"].item()) <= 3e-3 * f["loss"].item()
    assert losses[-1] < 0.5 * losses[0]


@pytest.mark.parametrize("shape", [
    (2, 64, 96, 12, 10),    # M = 240 < 1024: 128x128x32 base kernel, ragged last tile, non-square image
    (3, 32, 40, 20, 24),    # M = 1440: 256x128 phase kernel, Cout not a multiple of 32 (dX falls back to row2im)
    (2, 320, 320, 32, 32),  # UNet level (K = 2880 = 90 K-steps, 10 per tap); 24 tiles: the reduction is split 5 ways
    (1, 96, 64, 5, 7),      # image smaller than a tile row: every tile crosses image rows
    (2, 256, 128, 8, 8),    # the UNet's deepest level: one output tile, the reduction is split 4 ways (base kernel)
    (2, 64, 3, 20, 24),     # a thin head (the UNet's 3 output channels): zero filters up to 32 channels, all three passes implicit
    (3, 320, 5, 16, 16),    # the same from 320 channels, M = 768 (base kernel)
])
def test_implicit_conv3x3_matches_fp32_reference_and_im2row_route(shape):
    """cfhip_conv3x3_nhwc_bf16 / cfhip_conv3x3_wgrad_nhwc_bf16 (taps gathered inside the GEMM K loop, zero padding by per-lane range checks) against
    torch's fp32 conv2d on the bf16-rounded operands (CPU), forward / input gradient / weight + bias gradients, and
    against the im2row route of the same Function."""
    b, cin, cout, h, w = shape
    assert HF._thin_head_ok(cin, cout, 3, 3, 1, 1, 1, h, w) == (cout % 8 != 0)
    g = torch.Generator().manual_seed(b * 1000 + cin + h)
    x = bf16_round(torch.randn(b, cin, h, w, generator=g))
    wt = bf16_round(torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5))
    bias = torch.randn(cout, generator=g)
    dy = bf16_round(torch.randn(b, cout, h, w, generator=g))
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, br, stride=1, padding=1)
    yr.backward(dy)

    def run(implicit):
        HF.IMPLICIT_CONV = implicit
        try:
            xd = x.to(DEV).to(torch.bfloat16).requires_grad_(True)
            wd = wt.to(DEV).requires_grad_(True)
            bd = bias.to(DEV).requires_grad_(True)
            y = HF.conv2d(xd, wd, bd, 1, 1, 1)
            y.backward(dy.to(DEV).to(torch.bfloat16))
            return y, xd.grad, wd.grad, bd.grad
        finally:
            HF.IMPLICIT_CONV = True

    y, gx, gw, gb = run(True)
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (b, cout, h, w)
    assert_close(y, yr, 4e-3, "implicit conv forward")
    assert_close(gx, xr.grad, 4e-3, "implicit conv dX")
    assert_close(gw, wr.grad, 2e-3, "conv dW")
    assert_close(gb, br.grad, 1e-5, "conv db")
    y2, gx2, gw2, _ = run(False)
    # same bf16 operands, fp32 accumulation in a different k order: equal up to the last bf16 rounding
    assert_close(y, y2, 3e-3, "implicit vs im2row forward")
    assert_close(gx, gx2, 3e-3, "implicit vs im2row dX")
    assert_close(gw, gw2, 2e-5, "implicit vs im2row dW (fp32 sums in a different order)")
    # the zero padding is exact: a single bright pixel in a corner only reaches its 2x2 neighbourhood
    xi = torch.zeros(1, cin, h, w)
    xi[0, :, 0, 0] = 1.0
    yi = HF.conv2d(xi.to(DEV).to(torch.bfloat16), wt.to(DEV), None, 1, 1, 1).float().cpu()
    assert torch.count_nonzero(yi[0, :, 2:, :]) == 0 and torch.count_nonzero(yi[0, :, :, 2:]) == 0
    ref = torch.nn.functional.conv2d(xi, wt, None, stride=1, padding=1)
    assert_close(yi, ref, 4e-3, "corner impulse")


@pytest.mark.parametrize("form", [0, 1, 2, 3])
def test_conv3x3_every_tile_form_and_split_vs_fp32(form):
    """Every tile form of cfhip_conv3x3_nhwc_bf16 (csrc/gemm.hip conv_plan: 256x128x32 / 128x128x32 / the 80-column-per-wave 128x160
    tiles of round 6 with 32- and 64-channel K-steps — B image padded to whole DMA instructions, 16 x 80 epilogue strips handed out
    as a flat list of 16-byte pieces; form 3 falls back to 2 where Cin % 64 != 0),
    forced through the `conv_form` / `conv_split` options, against torch's fp32 conv2d on the same bf16 operands: ragged pixel counts,
    channel counts that end inside a tile / inside a wave's 80 columns, with and without bias, whole-K and split reductions (the
    split path writes f32 slabs through the same flat epilogue)."""
    shapes = [(1, 32, 320, 20, 24), (2, 64, 168, 13, 9), (1, 96, 488, 37, 21), (3, 32, 24, 5, 7), (1, 320, 640, 16, 16)]
    g = torch.Generator().manual_seed(form + 11)
    try:
        for b, cin, cout, h, w in shapes:
            x = bf16_round(torch.randn(b, cin, h, w, generator=g))
            wt = bf16_round(torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5))
            bias = torch.randn(cout, generator=g)
            ref = torch.nn.functional.conv2d(x, wt, bias, stride=1, padding=1).permute(0, 2, 3, 1).reshape(b * h * w, cout)
            x_rows = x.permute(0, 2, 3, 1).reshape(b * h * w, cin).contiguous().to(DEV).to(torch.bfloat16)
            wk = ops.conv3x3_pack_filters(wt.to(DEV).to(torch.bfloat16), False)
            ops.set_option("conv_form", form)
            outs = {}
            for split in (1, 3, -1):
                ops.set_option("conv_split", split)
                for bi in (bias.to(DEV), None):
                    y = ops.conv3x3_nhwc(x_rows, wk, bi, b, h, w)
                    want = ref if bi is not None else ref - bias
                    assert_close(y, want, 4e-3, f"form {form} split {split} {(b, cin, cout, h, w)} bias {bi is not None}")
                    outs[(split, bi is not None)] = y
            # whole-K results do not depend on the tile form: the K order inside a tile is the same in every form
            ops.set_option("conv_form", 1)
            ops.set_option("conv_split", 1)
            assert torch.equal(ops.conv3x3_nhwc(x_rows, wk, bias.to(DEV), b, h, w), outs[(1, True)])
    finally:
        ops.set_option("conv_form", -1)
        ops.set_option("conv_split", -1)


def test_conv3x3_filter_packing_bit_exact():
    """cfhip_conv3x3_pack_filters: pure data movement, compared bit for bit with the permutes it replaces."""
    g = torch.Generator().manual_seed(5)
    for cout, cin in ((40, 32), (320, 640), (8, 96)):
        w16 = torch.randn(cout, cin, 3, 3, generator=g).to(DEV).to(torch.bfloat16)
        wk = ops.conv3x3_pack_filters(w16, False)
        wr = ops.conv3x3_pack_filters(w16, True)
        assert torch.equal(wk, w16.permute(0, 2, 3, 1).reshape(cout, 9 * cin))
        assert torch.equal(wr, w16.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, 9 * cout))


@pytest.mark.parametrize("cin,cout,dtype", [(40, 24, torch.bfloat16), (64, 3, torch.float32), (320, 640, torch.bfloat16)])
def test_pointwise_convolution_route(cin, cout, dtype):
    """1x1 / stride 1 convolutions (the UNet's skip connections) take the transpose route of Conv2dFn: NHWC rows saved for the
    weight gradient, dX back through a batched transpose; Cout not a multiple of 8 is padded.  Against torch fp32 on the same
    bf16-rounded operands, two backward passes (accumulation onto existing gradients)."""
    torch.manual_seed(cin + cout)
    b, h, w = 3, 9, 12
    x = bf16_round(torch.randn(b, cin, h, w))
    wt = bf16_round(torch.randn(cout, cin, 1, 1) * 0.2)
    bias = torch.randn(cout) * 0.1
    gy = bf16_round(torch.randn(b, cout, h, w))
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    want = torch.nn.functional.conv2d(xr, wr, br)
    want.backward(gy)
    xd = x.to(DEV).to(dtype).requires_grad_(True)
    wd, bd = wt.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    for rep in (1, 2):
        y = HF.conv2d(xd, wd, bd, 1, 0, 1)
        y.backward(gy.to(DEV).bfloat16())
        assert_close(y, want, 6e-3, "1x1 conv y")
        assert_close(wd.grad, rep * wr.grad, 1e-2, "1x1 conv gw")
        assert_close(bd.grad, rep * br.grad, 6e-3, "1x1 conv gb")
    assert_close(xd.grad, 2 * xr.grad, 1e-2, "1x1 conv gx")


@pytest.mark.parametrize("b,cin,cout,h,w,k,s,p,d,g,bias", [
    (2, 16, 16, 14, 14, 3, 1, 1, 1, 16, True),     # depthwise (DepthWiseConv2d)
    (3, 32, 64, 9, 11, 3, 2, 1, 1, 4, True),       # strided, 8 -> 16 channels per group
    (2, 24, 24, 13, 10, 5, 1, 2, 1, 24, False),    # depthwise 5x5
    (1, 12, 6, 8, 8, 3, 1, 2, 2, 3, True),         # dilated
    (2, 8, 8, 7, 7, 7, 1, 3, 1, 2, True),          # 49 taps
    (2, 6, 10, 5, 6, 1, 1, 0, 1, 2, True),         # 1x1
])
def test_grouped_convolution_vs_torch(b, cin, cout, h, w, k, s, p, d, g, bias):
    """`Conv2d(groups > 1)` (reference convs/basic.py:160-177, `F.conv2d(..., groups=g)`): output, input gradient, filter and
    bias gradients against fp32 torch on the same bf16-rounded operands; second backward accumulates."""
    torch.manual_seed(b * 100 + cin + k)
    conv = C.modules.Conv2d(cin, cout, kernel_size=k, groups=g, stride=s, dilation=d, padding=p, bias=bias).to(DEV)
    assert tuple(conv.weight.shape) == (cout, cin // g, k, k)
    with torch.no_grad():
        conv.weight.mul_(3.0)
        if bias:
            conv.bias.normal_()
    x = bf16_round(torch.randn(b, cin, h, w))
    xd = x.to(DEV).requires_grad_(True)
    y = conv(xd)
    xr = x.clone().requires_grad_(True)
    wr = bf16_round(conv.weight.detach().cpu()).requires_grad_(True)
    br = conv.bias.detach().cpu().clone().requires_grad_(True) if bias else None
    want = torch.nn.functional.conv2d(xr, wr, br, stride=s, padding=p, dilation=d, groups=g)
    assert y.shape == want.shape
    assert_close(y, want.detach(), 5e-3, "grouped conv forward")
    gy = bf16_round(torch.randn(want.shape))
    want.backward(gy)
    y.backward(gy.to(DEV))
    assert_close(xd.grad, xr.grad, 5e-3, "dX")
    assert_close(conv.weight.grad, wr.grad, 1e-4, "dW")
    if bias:
        assert_close(conv.bias.grad, br.grad, 1e-4, "db")
    conv(xd).backward(gy.to(DEV))  # accumulation
    assert_close(conv.weight.grad, 2 * wr.grad, 1e-4, "dW accumulated")
    if bias:
        assert_close(conv.bias.grad, 2 * br.grad, 1e-4, "db accumulated")


def test_depthwise_module_state_keys_and_eval():
    m = C.modules.DepthWiseConv2d(32).to(DEV)
    assert sorted(m.state_dict()) == ["net.bias", "net.weight"] and tuple(m.net.weight.shape) == (32, 1, 3, 3)
    x = torch.randn(2, 32, 10, 10, device=DEV)
    with torch.no_grad():
        y = m(x)
    want = torch.nn.functional.conv2d(bf16_round(x.cpu()), bf16_round(m.net.weight.detach().cpu()), m.net.bias.detach().cpu(), padding=1, groups=32)
    assert_close(y, want, 5e-3, "depthwise")


def test_grouped_convolution_is_linear_and_block_diagonal():
    """properties that do not need a reference: conv(a x + b y) == a conv(x) + b conv(y) without bias, and a group's output
    channels do not see the other groups' input channels (zeroing them changes nothing)."""
    torch.manual_seed(9)
    conv = C.modules.Conv2d(32, 48, kernel_size=3, groups=4, padding=1, bias=False).to(DEV)
    x, y = bf16_round(torch.randn(2, 32, 12, 12)).to(DEV), bf16_round(torch.randn(2, 32, 12, 12)).to(DEV)
    with torch.no_grad():
        lhs = conv(bf16_round(2.0 * x.cpu() + 0.5 * y.cpu()).to(DEV)).float()
        rhs = 2.0 * conv(x).float() + 0.5 * conv(y).float()
        assert_close(lhs, rhs, 1.5e-2, "linearity (bf16 rounding of the combined input and of both outputs)")
        x2 = x.clone()
        x2[:, 8:] = 0  # groups 1..3 of the input
        assert torch.equal(conv(x)[:, :12], conv(x2)[:, :12])  # group 0's 12 output channels: bit-identical


# ---- round 5: the Conv2d options that used to raise (reference convs/basic.py:41-177) ----------------------------------------------


@pytest.mark.parametrize("shape,pads", [((2, 3, 9, 7), (1, 1, 1, 1)), ((1, 5, 6, 6), (3, 0, 2, 5)), ((2, 8, 4, 12), (0, 3, 3, 0)),
                                        ((1, 2, 2, 2), (1, 1, 1, 1)), ((3, 4, 17, 5), (4, 4, 4, 4))])
def test_reflect_pad2d_vs_torch(shape, pads):
    """cfhip_reflect_pad2d_fwd / _bwd behind `Conv2d(padding="reflection[N]")` (nn.ReflectionPad2d, basic.py:61-75): the forward is a
    copy (bit-equal to F.pad(mode="reflect") of the bf16-rounded input, from f32 and from bf16 sources); the backward gathers the <= 9
    mirrored positions — against torch's gradient; pads as large as the extent allows (size - 1), asymmetric, zero."""
    torch.manual_seed(sum(shape) + sum(pads))
    x = torch.randn(*shape)
    want = torch.nn.functional.pad(bf16_round(x), list(pads), mode="reflect")
    for src in (x, x.to(torch.bfloat16)):
        xd = src.to(DEV).requires_grad_(True)
        y = HF.reflect_pad2d(xd, pads)
        assert y.dtype == torch.bfloat16 and torch.equal(y.float().cpu(), want)
    g = bf16_round(torch.randn(want.shape))
    xr = bf16_round(x).requires_grad_(True)
    torch.nn.functional.pad(xr, list(pads), mode="reflect").backward(g)
    y.backward(g.to(DEV))
    assert_close(xd.grad, xr.grad, 4e-3, "reflection-pad backward")
    with pytest.raises(RuntimeError):
        HF.reflect_pad2d(x.to(DEV), (shape[3], 0, 0, 0))  # a pad must be smaller than the extent it mirrors


@pytest.mark.parametrize("b,cin,cout,h,w,k,s,p,d", [(2, 16, 8, 7, 7, 4, 2, 1, 1), (1, 8, 24, 5, 9, 3, 1, 1, 1), (2, 6, 5, 6, 6, 3, 2, 0, 1),
                                                      (1, 32, 16, 8, 8, 2, 2, 0, 1), (1, 8, 8, 6, 5, 3, 1, 2, 2)])
def test_conv_transpose2d_vs_torch(b, cin, cout, h, w, k, s, p, d):
    """`functional.ConvTranspose2dFn` (Conv2d.forward(transpose=True), basic.py:151-160 -> F.conv_transpose2d): output, input gradient
    and weight gradient against fp32 torch on the same bf16-rounded operands; the stride-2 4x4 GAN up-sampling form, channel counts
    that are no multiple of 8 (generic GEMM path), dilation."""
    torch.manual_seed(b + cin + cout + k)
    x = bf16_round(torch.randn(b, cin, h, w))
    wt = bf16_round(torch.randn(cin, cout, k, k) * 0.2)
    xd, wd = x.to(DEV).requires_grad_(True), wt.to(DEV).requires_grad_(True)
    y = HF.conv_transpose2d(xd, wd, s, p, d)
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    want = torch.nn.functional.conv_transpose2d(xr, wr, None, stride=s, padding=p, dilation=d)
    assert y.shape == want.shape, (y.shape, want.shape)
    assert_close(y, want.detach(), 5e-3, "transposed convolution forward")
    gy = bf16_round(torch.randn(want.shape))
    want.backward(gy)
    y.backward(gy.to(DEV))
    assert_close(xd.grad, xr.grad, 5e-3, "dX")
    assert_close(wd.grad, wr.grad, 2e-4, "dW")


@pytest.mark.parametrize("kw", [
    dict(kernel_size=3, padding="reflection"), dict(kernel_size=3, padding="reflection", transform_kernel=True),
    dict(kernel_size=3, demodulate=True, weight_scale=0.5), dict(kernel_size=3, bias=False, demodulate=True, style=True),
    dict(kernel_size=4, stride=2, padding=1, transpose=True),
])
def test_conv2d_round5_options_on_the_gpu(kw):
    """the module end to end on the HIP path: reflection padding (kernel), kernel transform / demodulation / scale / style (weight math
    in torch + the HIP convolution; `style` = one group per sample on the grouped kernels), transposed form — against the same
    composition of torch functionals in fp32 on bf16-rounded activations and effective weights."""
    import torch.nn.functional as F

    kw = dict(kw)
    style_on, transpose = kw.pop("style", False), kw.pop("transpose", False)
    torch.manual_seed(11)
    m = C.modules.Conv2d(8, 16, **kw).to(DEV)
    with torch.no_grad():
        m.weight.mul_(2.0)
        if m.bias is not None:
            m.bias.normal_()
    x = bf16_round(torch.randn(2, 8, 10, 10))
    style = torch.randn(2, 8) if style_on else None
    xd = x.to(DEV).requires_grad_(True)
    y = m(xd, None if style is None else style.to(DEV), transpose=transpose)
    # the reference composition on the CPU
    xr = x.clone().requires_grad_(True)
    net = xr if m.reflection_pad is None else F.pad(xr, list(m.reflection_pad), mode="reflect")
    mc = C.modules.Conv2d(8, 16, **kw)
    mc.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    w = mc._effective_weight(style)
    w = w.detach().to(torch.bfloat16).float()  # what the MFMA sees
    bias = None if mc.bias is None else mc.bias.detach()
    if style_on:
        want = F.conv2d(net.reshape(1, 16, 10, 10), w.reshape(32, 8, 3, 3), None, stride=mc.stride, padding=mc.padding, groups=2).reshape(2, 16, 10, 10)
    elif transpose:
        want = F.conv_transpose2d(net, w.transpose(0, 1), bias, stride=mc.stride, padding=mc.padding)
    else:
        want = F.conv2d(net, w, bias, stride=mc.stride, padding=mc.padding)
    assert y.shape == want.shape, (y.shape, want.shape)
    assert_close(y, want.detach(), 6e-3, "output")
    gy = bf16_round(torch.randn(want.shape))
    want.backward(gy)
    y.backward(gy.to(DEV))
    HF.SideStream.join()
    assert_close(xd.grad, xr.grad, 6e-3, "dX")
    assert m.weight.grad is not None and torch.isfinite(m.weight.grad).all() and float(m.weight.grad.abs().max()) > 0
