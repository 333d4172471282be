"""Host-side logic that needs no GPU: registry semantics, state_dict compatibility with the
reference (through the golden fixture keys), mask-layout quirk (bit-exact), arenas, split-K policy."""
import pytest
import torch

import cflearn_amd as C
import vit_oracle as O


def test_registry_build_module_drops_unknown_kwargs():
    @C.register_module("unit.dummy")
    class Dummy(torch.nn.Module):
        def __init__(self, a: int, b: int = 2):
            super().__init__()
            self.a, self.b = a, b

    m = C.build_module("unit.dummy", config=dict(a=1, zzz=3))
    assert (m.a, m.b) == (1, 2)
    m = C.build_module("unit.dummy", config=dict(a=1), b=5)
    assert m.b == 5
    with pytest.raises(TypeError):
        C.build_module("unit.dummy", config=dict(b=1))
    with pytest.raises(KeyError):
        C.build_module("unit.nope")
    pm = C.PrefixModules("unit")
    assert pm.has("dummy") and pm.get("dummy") is Dummy and "unit.dummy" in pm.all


def test_reference_names_registered():
    for name in ("attention.basic", "token_mixer.attention", "channel_mixer.ff", "encoders.vit", "cv_clf"):
        assert name in C.module_dict


def test_state_dict_keys_match_reference(golden):
    g = golden("vit_small.pt")
    cfg = dict(g["cfg"])
    m = C.build_module("cv_clf", config=dict(in_channels=3, num_classes=g["num_classes"], img_size=cfg.pop("img_size"),
                                             latent_dim=cfg["latent_dim"], encoder="vit", encoder_config=cfg))
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["sd"].keys())  # same names AND same registration order
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(g["sd"][k].shape), k
    m.load_state_dict(g["sd"])  # reference checkpoint loads as is


def test_vit_b16_parameter_count():
    m = C.vit_b16_classifier()
    assert sum(p.numel() for p in m.parameters()) == 86_567_656  # SURVEY §6 probe of the reference
    assert len(m.state_dict()) == 152
    att = m.encoder.encoder.mixing_blocks[0].token_mixing.net
    assert att.num_heads == 12 and att.head_dim == 64 and tuple(att.in_w.shape) == (2304, 768)
    assert m.encoder.encoder.mixing_blocks[0].token_norm.eps == 1.0e-6


def test_attention_module_parameter_layouts():
    a = C.Attention(128, 2, is_self_attention=True)
    assert set(dict(a.named_parameters())) == {"in_w", "qkv_bias", "out_linear.linear.weight", "out_linear.linear.bias"}
    a = C.Attention(128, 2, k_dim=64, v_dim=32)
    assert {"q_w", "k_w", "v_w", "qkv_bias"} <= set(dict(a.named_parameters()))
    a = C.Attention(128, 2)
    assert {"q_w", "kv_w", "q_bias", "kv_bias"} <= set(dict(a.named_parameters()))
    with pytest.raises(ValueError):
        C.Attention(100, 3)


def test_mask_quirk_is_bit_exact():
    torch.manual_seed(0)
    for b, h in ((3, 2), (4, 3), (2, 5)):
        mask = torch.rand(b, 7, 9) < 0.4
        keep = C.modules.expand_module_mask(mask, h)
        want = O.expand_module_mask(mask, h)
        assert keep.dtype == torch.uint8 and torch.equal(keep.bool(), want)
        # and it is what the reference's repeat/view produces (attentions.py:246-249)
        ref = (~mask).repeat(h, 1, 1).view(-1, h, 7, 9)
        assert torch.equal(keep.bool(), ref)


def test_unsupported_features_raise():
    # round 5: built — pruned weights, reflection padding, the StyleGAN weight forms, transposed convolution
    pr = C.Linear(8, 8, pruner_config={})
    assert isinstance(pr.pruner, C.modules.Pruner) and sorted(k for k in pr.state_dict() if k.startswith("pruner.")) == [
        "pruner.alpha", "pruner.beta", "pruner.eps", "pruner.gamma", "pruner.max_ratio"]
    with pytest.raises(NotImplementedError):
        C.Attention(128, 2, reduction_ratio=2)
    with pytest.raises(NotImplementedError):
        C.FeedForward(8, 16, 0.0, activation="glu")
    grouped = C.Conv2d(4, 8, kernel_size=3, groups=2)  # round 4: built (csrc/conv_grouped.hip); the reference's parameter shape
    assert tuple(grouped.weight.shape) == (8, 2, 3, 3) and grouped.groups == 2
    with pytest.raises(ValueError):
        C.Conv2d(6, 8, kernel_size=3, groups=4)
    assert C.Conv2d(4, 8, kernel_size=3, transform_kernel=True)._effective_weight(None).shape == (8, 4, 4, 4)  # one tap larger
    assert C.Conv2d(3, 8, kernel_size=3, padding="reflection").reflection_pad == (1, 1, 1, 1)
    assert C.Conv2d(3, 8, kernel_size=5, padding="reflection3", transform_kernel=True).reflection_pad == (4, 3, 4, 3)
    with pytest.raises(ValueError):
        C.Conv2d(3, 8, kernel_size=3, padding="circular")
    # round 2: dropout / DropPath are built (csrc/random.hip): the constructors keep the reference's modules in place
    mp = C.modules.Mapping(8, 8, dropout=0.5)
    assert isinstance(mp.dropout, C.modules.Dropout) and mp.dropout.p == 0.5
    assert C.modules.Mapping(8, 8, dropout=0.0).dropout is None  # mappings.py:66-67
    blk = C.MixingBlock(0, 2, 5, 64, 128, token_mixing_type="attention", token_mixing_config=dict(num_heads=1),
                        dropout=0.1, drop_path=0.2, norm_type="layer")
    assert isinstance(blk.drop_path, C.modules.DropPath) and blk.drop_path.dropout == 0.2
    assert blk._stochastic() and not blk._fusable()      # training mode: the composed path with the random masks
    blk.eval()
    assert not blk._stochastic()                          # eval mode: dropout is the identity, fused path allowed


def test_param_arena_views_and_lazy_zero():
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    before = [p.detach().clone() for p in lin.parameters()]
    arena = C.ParamArena(lin.parameters(), with_shadow=True)
    for p, b, off in zip(lin.parameters(), before, arena.offsets):
        assert torch.equal(p.detach(), b)
        assert off % 8 == 0
        assert p.data_ptr() == arena.flat_p.data_ptr() + 4 * off
        assert p.grad.data_ptr() == arena.flat_g.data_ptr() + 4 * off
        assert torch.equal(p._cfhip_shadow.float(), b.to(torch.bfloat16).float())
    # autograd accumulates in place into the arena views
    arena.zero_grad()
    lin(torch.randn(4, 5)).sum().backward()
    g0 = arena.flat_g.clone()
    assert g0.abs().sum() > 0
    # lazy zeroing: nothing written -> finalize zeroes the stale values
    arena.zero_grad(lazy=True)
    assert arena.flat_g.abs().sum() > 0
    arena.finalize_grads()
    assert arena.flat_g.abs().sum() == 0


def test_fused_adam_is_gpu_only():
    lin = torch.nn.Linear(4, 4)
    opt = C.FusedAdam(lin.parameters(), lr=1e-3)
    lin(torch.randn(2, 4)).sum().backward()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()


def test_split_k_policy():
    from cflearn_amd.ops import pick_split_k

    assert pick_split_k(12608, 3072, 768) == 1           # forward GEMM: plenty of tiles
    assert pick_split_k(768, 768, 12608) >= 8            # dW of a 768x768 layer: 36 tiles only
    s = pick_split_k(2304, 768, 12608)
    assert 2 <= s <= 8
    assert pick_split_k(1000, 768, 64) == 1
    assert pick_split_k(320, 320, 65536) == 32 and pick_split_k(640, 640, 16384) == 21 and pick_split_k(320, 2880, 32768) == 8


def test_conv_classifier_and_fcnn_state_dict_keys_match_reference(golden):
    """cv_clf(encoder="vanilla_1d") (examples/cv/classification/mnist_clf.py) and fcnn: key-for-key, shape-for-shape
    equal to the state_dict the reference's own modules produced (fixtures made by oracle/gen_golden.py)."""
    g = golden("mnist_clf.pt")
    m = C.build_module("cv_clf", config=dict(in_channels=1, num_classes=10, encoder_config=dict(num_downsample=3)))
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["sd"].keys())
    assert all(sd[k].shape == g["sd"][k].shape and sd[k].dtype == g["sd"][k].dtype for k in sd)
    m.load_state_dict(g["sd"])
    f = golden("fcnn.pt")
    n = C.build_module("fcnn", config=dict(input_dim=96, output_dim=10))
    assert list(n.state_dict().keys()) == list(f["sd"].keys())
    assert all(n.state_dict()[k].shape == f["sd"][k].shape for k in f["sd"])
    n.load_state_dict(f["sd"])
    assert n.hidden_units == [192, 192]


def test_clip_state_dict_keys_match_reference(golden):
    g = golden("clip_small.pt")
    m = C.build_module("clip", config=dict(g["cfg"]))
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["sd"].keys())
    assert all(sd[k].shape == g["sd"][k].shape and sd[k].dtype == g["sd"][k].dtype for k in sd)
    assert torch.equal(sd["text_transformer.attention_mask"], g["sd"]["text_transformer.attention_mask"])
    m.load_state_dict(g["sd"])


def test_ddpm_noise_schedule_matches_reference_bitwise(golden):
    """diffusion.NoiseSchedule restates make_beta_schedule("linear") + the q(x_t | x_0) tables in float64 numpy like the
    reference (ddpm.py:51-89,599-640): the fp32 buffers are bit-equal to the reference-made fixture."""
    from cflearn_amd.diffusion import NoiseSchedule

    g = golden("ddpm_schedule.pt")
    s = NoiseSchedule(1000, "linear", 8.5e-4, 1.2e-2)
    assert torch.equal(s.betas, g["betas"])
    assert torch.equal(s.sqrt_alphas_cumprod, g["sqrt_alphas_cumprod"])
    assert torch.equal(s.sqrt_one_minus_alphas_cumprod, g["sqrt_one_minus_alphas_cumprod"])


def test_adam_hyper_record_ring_host_side():
    """FusedAdam.prepare_step: every step fills ITS OWN slot of the pinned ring (a single host buffer was overwritten
    by a host running ahead of the GPU), and the record holds the bias corrections of that step."""
    import math

    import torch

    from cflearn_amd.optim import _HYPER_RING, FusedAdam, ParamArena

    p = torch.nn.Parameter(torch.zeros(16))
    opt = FusedAdam(None, lr=2e-3, betas=(0.9, 0.99), eps=1e-7, weight_decay=0.05, arena=ParamArena([p], with_shadow=False))
    opt.grad_scale = 0.5
    seen = []
    for t in range(1, 2 * _HYPER_RING + 2):
        opt.prepare_step()
        h = opt._hyper_dev.clone()
        want = [2e-3, 0.9, 0.99, 1e-7, 0.05, 1.0 - 0.9 ** t, 1.0 / math.sqrt(1.0 - 0.99 ** t), 0.5]
        assert torch.allclose(h, torch.tensor(want, dtype=torch.float32), rtol=1e-6, atol=0), (t, h)
        seen.append(opt._hyper_host.data_ptr())
    assert len(set(seen[:_HYPER_RING])) == _HYPER_RING           # distinct slots within one lap
    assert seen[:_HYPER_RING] == seen[_HYPER_RING:2 * _HYPER_RING]  # and the ring wraps


def test_gradients_return_to_the_arena_after_set_to_none():
    """The reference trainer calls `optimizer.zero_grad()` (torch default: set_to_none=True) every step.  A direct-writing
    backward (functional.write_param_grad) and the reducer's adoption of autograd-made gradients must both end up in
    the arena slot, which is what the bucketed all-reduce and the fused Adam read."""
    import torch

    from cflearn_amd import functional as HF

    lin = torch.nn.Linear(3, 2)
    arena = C.ParamArena(lin.parameters(), with_shadow=False)
    w, b = lin.weight, lin.bias
    torch.optim.SGD(lin.parameters(), lr=0.1).zero_grad()  # set_to_none
    assert w.grad is None and b.grad is None
    arena.flat_g.fill_(7.0)  # stale values of the previous step
    HF.write_param_grad(w, lambda out, acc: out.copy_(torch.full_like(out, 2.0)) if not acc else out.add_(2.0))
    assert w.grad.data_ptr() == arena.grad_view(w).data_ptr() and torch.all(w.grad == 2.0)
    HF.write_param_grad(w, lambda out, acc: out.copy_(torch.full_like(out, 2.0)) if not acc else out.add_(2.0))
    assert torch.all(arena.grad_view(w) == 4.0)  # the second contribution of the same backward accumulates
    # autograd-made gradient of the bias: adopted into its slot
    (lin(torch.ones(1, 3)).sum()).backward(inputs=[b])
    assert b.grad.data_ptr() != arena.grad_view(b).data_ptr()
    arena.adopt_grad(b)
    assert b.grad.data_ptr() == arena.grad_view(b).data_ptr() and torch.all(b.grad == 1.0)
    # a parameter outside any arena still gets a private tensor
    q = torch.nn.Parameter(torch.zeros(4))
    HF.write_param_grad(q, lambda out, acc: out.copy_(torch.ones_like(out)))
    assert q.grad is not None and torch.all(q.grad == 1.0)


def test_fused_adam_behind_the_torch_optimizer_interface():
    """FusedAdamOptimizer: what the reference's scheduler / accelerate / DDP-callback code needs from an optimizer."""
    import torch

    from cflearn_amd.optim import FusedAdamOptimizer, FusedAdamWOptimizer

    lin = torch.nn.Linear(4, 3)
    opt = FusedAdamWOptimizer(lin.parameters(), lr=2e-3)
    assert isinstance(opt, torch.optim.Optimizer) and opt.fused.decoupled and opt.defaults["weight_decay"] == 1e-2
    assert not FusedAdamOptimizer(torch.nn.Linear(2, 2).parameters()).fused.decoupled
    # parameters and gradients live in the arena, zero_grad keeps them bound
    for p in lin.parameters():
        assert p.grad is not None and p.grad.data_ptr() == opt.arena.grad_view(p).data_ptr()
    lin(torch.randn(5, 4)).sum().backward()
    assert opt.arena.flat_g.abs().sum() > 0
    opt.zero_grad()
    assert opt.arena.flat_g.abs().sum() == 0 and lin.weight.grad is not None
    # a torch scheduler (the reference's schedulers subclass _LRScheduler) drives the lr the kernel will read
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda epoch: 0.5 ** epoch)
    assert opt.fused.param_groups is opt.param_groups
    opt.fused.prepare_step()
    assert abs(opt.fused._hyper_dev[0].item() - 2e-3) < 1e-9
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "scheduler.step() before optimizer.step()": no GPU here to step with
        sched.step()
    opt.fused.prepare_step()
    assert abs(opt.fused._hyper_dev[0].item() - 1e-3) < 1e-9
    # step hooks (ddp.RcclDDPCallback registers one) fire; the update itself is GPU-only
    seen = []
    opt.register_step_pre_hook(lambda *a, **k: seen.append(1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()
    assert seen == [1]
    sd = opt.state_dict()
    assert sd["step"] == opt.fused.step_count and sd["exp_avg"].shape == opt.arena.flat_p.shape
    with pytest.raises(ValueError, match="one parameter group"):
        FusedAdamOptimizer([dict(params=[torch.nn.Parameter(torch.zeros(2))]), dict(params=[torch.nn.Parameter(torch.zeros(2))])])


def test_lazy_zero_grad_with_gradients_that_arrive_through_autograd():
    """`ParamArena.zero_grad(lazy=True)` marks every slot 'fresh' instead of clearing it.  A gradient that autograd
    accumulates itself (the op saw a slice of the parameter) must neither be added onto the previous step's values nor be
    wiped by `finalize_grads()`: the arena's tensor hook clears the slot in front of the accumulation."""
    from cflearn_amd.functional import write_param_grad
    from cflearn_amd.optim import ParamArena

    torch.manual_seed(0)
    a = torch.nn.Parameter(torch.randn(6, 4))   # gets its gradient through autograd (sliced use)
    b = torch.nn.Parameter(torch.randn(5))      # direct-written
    c = torch.nn.Parameter(torch.randn(3))      # unused: must end up zero
    arena = ParamArena([a, b, c], with_shadow=False)
    for step in range(3):
        arena.zero_grad(lazy=True)
        x = torch.randn(4)
        loss = (a[1:4] @ x).sum() * (step + 1) + (a[4:] ** 2).sum()
        loss.backward()
        write_param_grad(b, lambda out, acc: out.add_(torch.ones(5)) if acc else out.copy_(torch.full((5,), 2.0 + step)))
        arena.finalize_grads()
        want = torch.zeros(6, 4)
        want[1:4] = x * (step + 1)
        want[4:] = 2 * a.detach()[4:]
        assert torch.allclose(a.grad, want), step
        assert torch.equal(b.grad, torch.full((5,), 2.0 + step))
        assert torch.equal(c.grad, torch.zeros(3))
        assert a.grad.data_ptr() == arena.grad_view(a).data_ptr()


def test_unet_variant_state_dict_keys_and_shapes_match_reference(golden):
    """Every UNet option of round 2 builds the reference's parameter layout (names, order, shapes): reference-made
    checkpoints load — `MultiHeadSpatialAttention` (both head layouts), the scale-shift residual block (time embedding of
    2 x out_channels), the class-conditional pixel-attention UNet with ResBlock resampling, the Linear-projection
    spatial-transformer UNet."""
    from cflearn_amd.modules import MultiHeadSpatialAttention, ResidualBlockWithTimeEmbedding

    g = golden("unet_variants.pt")

    def same(m, sd):
        mine = m.state_dict()
        assert list(mine.keys()) == list(sd.keys())
        for k in sd:
            assert tuple(mine[k].shape) == tuple(sd[k].shape), k
        m.load_state_dict(sd)

    for c in g["mhsa"]:
        same(MultiHeadSpatialAttention(**c["cfg"]), c["sd"])
    same(ResidualBlockWithTimeEmbedding(**g["scale_shift"]["cfg"]), g["scale_shift"]["sd"])
    for c in g["unets"]:
        m = C.build_module("unet_diffuser", config=dict(c["cfg"]))
        same(m, c["sd"])
        assert sum(p.numel() for p in m.parameters()) == sum(v.numel() for v in c["sd"].values())


def test_bench_gemm_shapes_conserve_flops_under_forward_slices():
    """bench.gemm_shapes lists the launches of one step; slicing the forward changes the launch shapes, not the work"""
    import bench
    from cflearn_amd import fused

    keep = fused.FWD_HALVES, fused.BWD_HALVES, fused.DW_GROUP_BLOCKS

    def flops(row):
        c, lay, m, n, k, _ = row
        if lay == "tn-grouped":  # m = ((M, N, K), ...) of one grouped weight-gradient launch
            return c * sum(2.0 * mm * nn * kk for mm, nn, kk in m)
        return c * 2.0 * m * n * k

    try:
        totals = []
        for v, bv, grp in ((1, 1, 0), (2, 1, 0), (3, 2, 2), (2, 2, 1), (2, 3, 5), (1, 2, 12)):
            fused.FWD_HALVES, fused.BWD_HALVES, fused.DW_GROUP_BLOCKS = v, bv, grp
            shapes = bench.gemm_shapes(128)
            totals.append(sum(flops(r) for r in shapes))
            assert sum(c * m for c, lay, m, n, k, e in shapes if lay == "nt" and e == "gelu") == 12 * 128 * 197
            assert sum(c * m for c, lay, m, n, k, e in shapes if lay == "nn" and e == "dgelu") == 12 * 128 * 197
            assert sum(c * len(m) for c, lay, m, *_ in shapes if lay == "tn-grouped") == (48 if grp else 0)
        assert all(tot == 12910053556224.0 for tot in totals), totals
    finally:
        fused.FWD_HALVES, fused.BWD_HALVES, fused.DW_GROUP_BLOCKS = keep


def test_scale_shift_affine_function_matches_autograd():
    """`functional.ScaleShiftAffineFn` (the per-sample affine the scale-shift GroupNorm kernel is fed, residual.py:236-239):
    outputs and all four gradients against plain autograd; the parameter gradients follow the direct-write protocol."""
    from cflearn_amd import functional as HF

    torch.manual_seed(0)
    b, c = 3, 8
    gamma, beta = torch.randn(c, requires_grad=True), torch.randn(c, requires_grad=True)
    scale, shift = torch.randn(b, c, requires_grad=True), torch.randn(b, c, requires_grad=True)
    ge, be = HF.scale_shift_affine(gamma, beta, scale, shift)
    wg, wb = torch.randn(b, c), torch.randn(b, c)
    ((ge * wg).sum() + (be * wb).sum()).backward()
    g2, b2, s2, t2 = (t.detach().clone().requires_grad_(True) for t in (gamma, beta, scale, shift))
    ge2 = g2[None] * (1 + s2)
    be2 = b2[None] * (1 + s2) + t2
    ((ge2 * wg).sum() + (be2 * wb).sum()).backward()
    assert torch.allclose(ge, ge2.detach()) and torch.allclose(be, be2.detach())
    for mine, ref in ((gamma, g2), (beta, b2), (scale, s2), (shift, t2)):
        assert torch.allclose(mine.grad, ref.grad, atol=1e-6), (mine.grad, ref.grad)
    # second use accumulates into the same .grad (not 'fresh'): twice the gradient
    ge, be = HF.scale_shift_affine(gamma, beta, scale.detach(), shift.detach())
    ((ge * wg).sum() + (be * wb).sum()).backward()
    assert torch.allclose(gamma.grad, 2 * g2.grad, atol=1e-6) and torch.allclose(beta.grad, 2 * b2.grad, atol=1e-6)


def test_attention_dropout_counter_accounting():
    """every attention call with dropout reserves exactly its B * H * ceil(Tq/4) * ceil(Tk/4) Philox counters, so no two
    calls (and no call and its neighbours in the stream) share random bits"""
    from cflearn_amd import functional as HF, ops

    ops.PhiloxState.manual_seed(11)
    kw1 = HF._take_attn_dropout(0.1, 2, 3, 197, 197)
    kw2 = HF._take_attn_dropout(0.25, 1, 1, 5, 9)
    assert kw1["offset"] == 0 and kw1["seed"] == 11 and kw1["dropout_p"] == 0.1
    assert kw2["offset"] == ops.attn_dropout_blocks(2, 3, 197, 197) == 2 * 3 * 50 * 50
    assert ops.PhiloxState.offset == kw2["offset"] + 1 * 1 * 2 * 3
    assert HF._take_attn_dropout(0.0, 2, 3, 4, 4) == {} and HF._take_attn_dropout(1.0, 2, 3, 4, 4) == {}


def test_whole_param_resolves_reshaped_views_only():
    """`functional.whole_param`: a contiguous view of a WHOLE leaf parameter (a 1x1 filter seen as its GEMM matrix) is
    written through its base parameter; slices, permutations and non-leaf tensors keep the autograd route."""
    from cflearn_amd.functional import whole_param

    w = torch.nn.Parameter(torch.randn(6, 4, 1, 1))
    assert whole_param(w) is w
    assert whole_param(w.view(6, 4)) is w
    assert whole_param(w.view(6, -1)) is w
    sl = w.view(6, 4)[2:5]
    assert whole_param(sl) is sl
    pm = w.view(2, 3, 4).transpose(0, 1).reshape(6, 4)  # a copy: not a view of the storage in order
    assert whole_param(pm) is pm
    frozen = torch.nn.Parameter(torch.randn(3, 3), requires_grad=False)
    v = frozen.view(9)
    assert whole_param(v) is v
    assert whole_param(None) is None
    h = torch.randn(4, 4)  # a non-parameter leaf that does not require grad
    assert whole_param(h.view(16)) is not h


def test_bench_self_launch_command_and_percentiles():
    """`python bench.py --gpus N` from a plain shell starts its own ranks: the launcher command line, the
    refusal when the node has fewer GPUs than ranks, and the percentile helper of the per-step timings."""
    import bench

    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], 29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert 1024 < bench.free_port() < 65536
    assert bench.self_launch(2, ["--gpus", "2"]) == 2  # no GPU in this container: refused, not launched
    vals = sorted([20.0, 21.0, 19.0, 25.0, 20.5])
    assert bench._pct(vals, 0.5) == 20.5 and bench._pct(vals, 0.0) == 19.0 and bench._pct(vals, 1.0) == 25.0
    assert abs(bench._pct(vals, 0.1) - 19.4) < 1e-9


def test_round3_host_rules():
    """Host-side rules added in round 3, no device needed: GroupNorm slice counts, batch-slice cuts, split-K cap, and the
    adjacency test that lets a self-attention run to_q | to_k | to_v as one GEMM."""
    import torch

    import cflearn_amd as C
    from cflearn_amd import functional as HF
    from cflearn_amd import fused, ops
    from cflearn_amd.modules import CrossAttention

    # GroupNorm: few samples -> slices; never slices shorter than GN_MIN_SLICE, never more than 64, 1 when B * G fills the chip
    assert ops.gn_splits(1, 320, 32, 256 * 256) == 32
    assert ops.gn_splits(8, 320, 32, 64 * 64) == 2
    assert ops.gn_splits(8, 1280, 32, 8 * 8) == 1
    assert ops.gn_splits(64, 320, 32, 64 * 64) == 1
    assert ops.gn_splits(1, 320, 32, 1001) == 1  # inner not a multiple of 8: the vector kernels do not apply
    assert ops.gn_splits(1, 32, 32, 1 << 24) == 32  # (want = 1024 / 32 workgroups)
    # batch slices: equal halves by default, the share knob moves the cut, one sample per slice at least
    assert fused._slice_cuts(128, 2) == [0, 64, 128] and fused._slice_cuts(7, 3) == [0, 2, 4, 7]
    keep = fused.FIRST_SLICE_SHARE
    try:
        fused.FIRST_SLICE_SHARE = 60 / 128
        assert fused._slice_cuts(128, 2) == [0, 60, 128]
        fused.FIRST_SLICE_SHARE = 0.0
        assert fused._slice_cuts(128, 2) == [0, 1, 128]
    finally:
        fused.FIRST_SLICE_SHARE = keep
    # split-K: capped at 32 slices, none for short reductions or outputs that fill the chip
    assert ops.pick_split_k(320, 320, 32768) == 32 and ops.pick_split_k(1280, 1280, 512) == 1
    assert ops.pick_split_k(4096, 4096, 100000) == 1 and 1 < ops.pick_split_k(768, 768, 25088) <= 32
    # adjacency of the three projection weights: true inside one arena in registration order, false without an arena, across
    # arenas, or when the shapes differ (a cross attention with its own context width)
    torch.manual_seed(0)
    m = CrossAttention(query_dim=64, num_heads=2, head_dim=16)
    assert not HF.qkv_weights_adjacent(m.to_q.weight, m.to_k.weight, m.to_v.weight)
    C.ParamArena(list(m.parameters()), with_shadow=True)
    assert HF.qkv_weights_adjacent(m.to_q.weight, m.to_k.weight, m.to_v.weight)
    assert not HF.qkv_weights_adjacent(m.to_k.weight, m.to_q.weight, m.to_v.weight)
    x = CrossAttention(query_dim=64, context_dim=48, num_heads=2, head_dim=16)
    C.ParamArena(list(x.parameters()), with_shadow=True)
    assert not HF.qkv_weights_adjacent(x.to_q.weight, x.to_k.weight, x.to_v.weight)
    a, b = CrossAttention(query_dim=64, num_heads=2, head_dim=16), CrossAttention(query_dim=64, num_heads=2, head_dim=16)
    C.ParamArena(list(a.parameters()), with_shadow=True)
    C.ParamArena(list(b.parameters()), with_shadow=True)
    assert not HF.qkv_weights_adjacent(a.to_q.weight, b.to_k.weight, a.to_v.weight)
    assert m._plain_projections()


def test_round4_host_rules(tmp_path):
    """Host-side rules added in round 4, no device needed: the tile form / tile count of a weight-gradient flush, the address
    fingerprint that keys a launch plan, the kernel-trace step marks of the ViT and the UNet, the vectorcall generator's
    coverage rule."""
    import gzip
    import os
    import subprocess
    import sys

    import torch

    from cflearn_amd import fused, ops

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # weight-gradient queue: counted in 256 x 256 output tiles
    dy, x = torch.empty(10, 768), torch.empty(10, 3072)
    w = torch.empty(768, 3072)
    assert fused._tiles_of([(w, None, dy, x)]) == 3 * 12

    # plan key: state + where parameters and gradients live
    p1, p2 = torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(4))
    assert fused._grad_state((p1, None, p2)) is None  # no gradients yet
    p1.grad, p2.grad = torch.zeros(4), torch.zeros(4)
    s0 = fused._grad_state((p1, None, p2))
    assert s0 is not None and s0[0] == 0
    p1._cfhip_fresh = p2._cfhip_fresh = True
    s1 = fused._grad_state((p1, None, p2))
    assert s1[0] == 1 and s1[1] == s0[1]
    p2.grad = torch.zeros(4)  # a replaced gradient buffer: another fingerprint
    assert fused._grad_state((p1, None, p2))[1] != s1[1]
    p2._cfhip_fresh = False
    assert fused._grad_state((p1, None, p2)) is None  # some written first, others accumulated: no plan

    # timeline step marks: the ViT's patch-embedding im2row, the UNet's diffusion-loss kernel (its conv_im2row launches are not marks)
    def trace(names):
        pth = tmp_path / "t.csv.gz"
        with gzip.open(pth, "wt") as f:
            f.write("queue,stream,start_us,dur_us,name\n")
            for i, n in enumerate(names):
                f.write(f"1,0,{i * 10.0:.2f},5.00,{n}\n")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "timeline_report.py"), str(pth)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return r.stdout.splitlines()[0]

    vit = ["(anonymous namespace)::im2row_kernel<false>", "k1", "softmax_xent_kernel", "k2"] * 3
    assert trace(vit).startswith("step window 40.0 us, 4 kernels")
    unet = ["(anonymous namespace)::conv_im2row_kernel<false>", "a", "(anonymous namespace)::conv_im2row_kernel<false>", "b",
            "diffusion_loss_kernel", "c"] * 3
    assert trace(unet).startswith("step window 60.0 us, 6 kernels")

    # vectorcall generator: every entry point whose arguments are plain numbers is wrapped, the others are named in its skip list
    out = tmp_path / "fastcall_gen.c"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "carefree-learn_amd", "csrc", "gen_fastcall.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and "wrappers" in r.stdout, r.stderr
    src = out.read_text()
    assert "w_cfhip_gemm_bf16(" in src and "w_cfhip_adam_step_dev(" in src and "w_cfhip_comm_init(" not in src
    assert src.count("METH_FASTCALL") >= 75 and "if (PyErr_Occurred()) return NULL;" in src


def test_taped_node_engine_against_autograd():
    """functional.run_taped (round 5: a sub-module as ONE autograd node) with Functions whose arithmetic runs on the CPU: the walk
    over the tape gives autograd's gradients — fan-in, a frozen parameter, a contiguous view of a whole parameter, a gradient
    returned in another dtype, the `dx_add` fold of a pending gradient — and says so when it cannot (a view of a taped tensor
    between two Functions, a plain torch op behind the last one, a second backward, a Function that re-enters autograd)."""
    from torch.autograd import Function
    import cflearn_amd.functional as HF

    class Scale(Function):
        @staticmethod
        def forward(ctx, x, w, k):
            ctx.save_for_backward(x, w)
            ctx.k = k
            return x * w * k

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            return ((g * w * ctx.k if ctx.needs_input_grad[0] else None),
                    ((g * x * ctx.k).sum(0) if ctx.needs_input_grad[1] else None), None)

    seen = []

    class Squash(Function):
        folds_dx_add = True

        @staticmethod
        def forward(ctx, x):
            ctx.save_for_backward(x)
            return x.tanh()

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            dx = g * (1 - x.tanh() ** 2)
            add = getattr(ctx, "dx_add", None)
            seen.append(add is not None)
            return dx if add is None else dx + add

    class Mat(Function):
        @staticmethod
        def forward(ctx, x, w2d):
            ctx.save_for_backward(x, w2d)
            return x @ w2d.t()

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            return g @ w, (g.t() @ x).double()  # the engine casts it like autograd's

    class Add2(Function):
        @staticmethod
        def forward(ctx, a, b):
            return a + b

        @staticmethod
        def backward(ctx, g):
            return g, g

    def body(x, t, w1, w2, w4):
        h = HF._apply(Scale, x, w1, 2.0)
        r = HF._apply(Scale, HF._apply(Squash, h), w2, 0.5)
        m = HF._apply(Mat, HF._apply(Add2, h, r), w4.view(4, 4))  # h fans out; w4 is a [4, 4, 1, 1] filter seen as its matrix
        return HF._apply(Add2, m, HF._apply(Scale, t, w1, 1.0))

    def leaves(dtype=torch.float32):
        torch.manual_seed(0)
        return [torch.randn(3, 4).to(dtype).requires_grad_(True), torch.randn(3, 4).to(dtype).requires_grad_(True),
                torch.randn(4).to(dtype).requires_grad_(True), torch.randn(4).to(dtype).requires_grad_(True),
                torch.randn(4, 4, 1, 1).to(dtype).requires_grad_(True)]

    a = leaves()
    ya = body(*a)
    ya.square().sum().backward()
    b = leaves()
    yb = HF.run_taped(lambda: body(*b), tuple(b))
    assert type(yb.grad_fn).__name__ == "TapedFnBackward" and torch.equal(ya, yb)
    yb.square().sum().backward()
    for p, q in zip(a, b):
        assert q.grad is not None and q.grad.dtype == q.dtype and torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-6)
    assert seen == [False, False]  # f32 gradients are not folded (the kernel operand is bf16)
    # bf16: the gradient h already has from the residual add rides into Squash's backward
    del seen[:]
    a, b = leaves(torch.bfloat16), leaves(torch.bfloat16)
    body(*a).float().sum().backward()
    HF.run_taped(lambda: body(*b), tuple(b)).float().sum().backward()
    assert seen == [False, True]
    for p, q in zip(a, b):
        assert torch.allclose(p.grad.float(), q.grad.float(), rtol=3e-2, atol=3e-2)
    # a frozen parameter
    c = leaves()
    c[3].requires_grad_(False)
    HF.run_taped(lambda: body(*c), tuple(c)).sum().backward()
    assert c[3].grad is None and c[2].grad is not None and c[4].grad is not None
    # a view of a taped tensor between two Functions
    d = leaves()
    with pytest.raises(HF.TapeBreak, match="view of a taped tensor"):
        HF.run_taped(lambda: HF._apply(Squash, HF._apply(Scale, d[0], d[2], 2.0).t()), tuple(d))
    assert getattr(HF._TAPE, "tape", None) is None
    with pytest.raises(HF.TapeBreak, match="not the result of a taped Function"):
        HF.run_taped(lambda: HF._apply(Squash, d[0]) * 2, tuple(d))  # a plain torch op at the end
    with pytest.raises(HF.TapeBreak, match="cannot run inside"):
        HF.run_taped(lambda: HF._apply(HF._CheckpointFn, lambda v: v * 2, 1, d[0]), tuple(d))
    # differentiated twice
    e = leaves()
    loss = HF.run_taped(lambda: body(*e), tuple(e)).sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="differentiated twice"):
        loss.backward()
    # without grad mode: nothing is taped
    with torch.no_grad():
        assert HF.run_taped(lambda: body(*a), tuple(a)).grad_fn is None


def test_groupnorm_nhwc_form_rule():
    """ops.gn_nhwc_splits: the group form (0) while B * G workgroups fill the chip AND a thread's rows fit the registers it caches them
    in; the slice form beyond that (the zoo UNet's 64^2 x 960 / 640 and 32^2 x 1920 levels) and for few samples (256^2 x 1)"""
    from cflearn_amd import ops

    assert ops.gn_nhwc_splits(8, 1024, 640, 32) == 0 and ops.gn_nhwc_splits(8, 1024, 320, 32) == 0  # 41 / 21 rows per thread
    assert ops.gn_nhwc_splits(8, 256, 2560, 32) == 0 and ops.gn_nhwc_splits(8, 64, 1280, 32) == 0
    assert ops.gn_nhwc_splits(8, 4096, 960, 32) == 32 and ops.gn_nhwc_splits(8, 4096, 640, 32, True) == 64  # 241 / 164
    assert ops.gn_nhwc_splits(8, 4096, 320, 32, True) == 64 and ops.gn_nhwc_splits(8, 1024, 1920, 32) == 32  # 81 / 128
    assert ops.gn_nhwc_splits(8, 1024, 1280, 32) == 0 and ops.gn_nhwc_splits(8, 1024, 960, 32) == 32  # 21 (16-byte accesses) / 61
    assert ops.gn_nhwc_splits(1, 65536, 320, 32) == 1024  # one sample: slices over the whole chip
    assert ops.gn_nhwc_splits(8, 4096, 96, 32) > 0  # 3 channels per group: no dword pairs
