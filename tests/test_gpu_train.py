"""End-to-end optimisation steps on the MI355X: arena + fused Adam + (optional) hipGraph replay."""
import math

import pytest
import torch

import vit_oracle as O
from helpers import assert_close

pytestmark = pytest.mark.gpu

import cflearn_amd as C  # noqa: E402
from cflearn_amd.engine import TrainStep  # noqa: E402
from cflearn_amd import functional as HF, ops  # noqa: E402

DEV = "cuda"


def _small(golden):
    g = golden("vit_small.pt")
    cfg = dict(g["cfg"])
    m = C.build_module("cv_clf", config=dict(in_channels=3, num_classes=g["num_classes"], img_size=cfg.pop("img_size"),
                                             latent_dim=cfg["latent_dim"], encoder="vit", encoder_config=cfg))
    m.load_state_dict(g["sd"])
    return g, m.to(DEV)


def test_one_adamw_step_matches_oracle(golden):
    """forward + CE + backward + AdamW on the HIP path vs the same step done by the CPU oracle: the loss against the
    reference fixture, and the UPDATE against the oracle's AdamW applied to the gradients this very backward produced
    (element for element, <= 2e-3 of the learning rate — a wrong bias correction, weight decay or moment update would
    be off by orders of magnitude more)."""
    from cflearn_amd.functional import SideStream

    g, m = _small(golden)
    lr, wd = 1e-3, 0.01
    ts = TrainStep(m, lr=lr, weight_decay=wd, decoupled=True, step_in_backward=False)  # (the gradients are read back below)
    img, labels = g["img"].to(DEV), g["labels"].view(-1).to(DEV)
    for step in (1, 2):  # the second step exercises non-trivial moments and bias corrections
        ts.optimizer.prepare_step()
        ts.optimizer.zero_grad()
        logits = m(img)["predictions"]
        loss, dl = C.ops.softmax_xent(logits, labels, 0.25)
        logits.backward(dl)
        SideStream.join()
        ts.arena.finalize_grads()
        if step == 1:
            assert abs(loss.item() / 4 - g["loss"].item()) <= 3e-3 * g["loss"].item()
        p0, g0 = ts.arena.flat_p.clone().cpu(), ts.arena.flat_g.clone().cpu()
        m0, v0 = ts.optimizer.exp_avg.clone().cpu(), ts.optimizer.exp_avg_sq.clone().cpu()
        ts.optimizer.launch_step()
        q = p0.clone()
        O.adamw_step(q, g0, m0, v0, step, lr, weight_decay=wd, decoupled=True)
        got = ts.arena.flat_p.cpu()
        assert (got - q).abs().max().item() <= 2e-3 * lr, (step, (got - q).abs().max().item())
        assert (got - p0).abs().max().item() > 0.5 * lr  # and it did move
        assert (ts.optimizer.exp_avg.cpu() - m0).abs().max().item() <= 1e-6 * max(1.0, m0.abs().max().item())
    for p in ts.arena.params:
        assert torch.equal(p._cfhip_shadow.cpu(), p.detach().cpu().to(torch.bfloat16))


@pytest.mark.parametrize("range_bytes", [64 << 10, 1 << 20, 32 << 20])
def test_optimizer_step_inside_backward_is_bit_identical(golden, range_bytes):
    """`optim.StepInBackward`: the arena ranges are updated on a side stream as soon as their gradients are final, while
    backward is still running.  After every step: parameters, moments and the (swapped-in) bf16 shadows are bit-equal to ONE
    launch of the same kernel over the whole arena on the state before the step and the gradients this backward left in the
    arena (a range updated before its last gradient kernel had finished, or twice, cannot pass), ranges really were launched
    from inside backward, and the trajectory follows the end-of-step engine."""
    from cflearn_amd import _lib

    g, m1 = _small(golden)
    _, m2 = _small(golden)
    img, labels = g["img"].to(DEV), g["labels"].view(-1).to(DEV)
    ref = TrainStep(m1, lr=2e-3, weight_decay=0.01, step_in_backward=False)
    ovl = TrainStep(m2, lr=2e-3, weight_decay=0.01, step_in_backward=True, range_bytes=range_bytes)
    assert ref.optimizer.in_backward is None and ovl.optimizer.in_backward is not None
    opt, ar = ovl.optimizer, ovl.arena
    for step in range(4):
        p0, m0, v0 = ar.flat_p.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone()
        l_ref = ref.step(img, labels).item()
        l_ovl = ovl.step(img, labels).item()
        torch.cuda.synchronize()
        # (LayerNorm parameter gradients of this 128-wide model come from the atomic round-1 kernel: not bitwise reproducible
        # between two runs, so the reference update is applied to the gradients of THIS run)
        p16 = torch.empty(ar.total, dtype=torch.bfloat16, device=DEV)
        rc = _lib.load().cfhip_adam_step_dev(p0.data_ptr(), ar.flat_g.data_ptr(), m0.data_ptr(), v0.data_ptr(), p16.data_ptr(), ar.total,
                                             opt._hyper_dev.data_ptr(), int(opt.decoupled), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "adam_step_dev")
        torch.cuda.synchronize()
        assert torch.equal(p0, ar.flat_p), step
        assert torch.equal(m0, opt.exp_avg) and torch.equal(v0, opt.exp_avg_sq), step
        assert torch.equal(p16, ar.flat_p16), step  # the shadow arena that is current after the swap
        for p in ar.params:
            assert ar.flat_p16.data_ptr() <= p._cfhip_shadow.data_ptr() < ar.flat_p16.data_ptr() + 2 * ar.total
            assert torch.equal(p._cfhip_shadow, p.detach().to(torch.bfloat16))
        if range_bytes < (32 << 20):
            assert opt.in_backward.launched_in_backward >= 2, opt.in_backward.launched_in_backward
        assert not opt._done
        assert abs(l_ref - l_ovl) <= 1e-4 * abs(l_ref), (step, l_ref, l_ovl)
        assert (ref.arena.flat_p - ar.flat_p).abs().max().item() <= 1e-5  # same trajectory (up to those LayerNorm atomics)


def test_training_reduces_loss_and_graph_matches_eager(golden):
    g, m1 = _small(golden)
    _, m2 = _small(golden)
    img, labels = g["img"].to(DEV), g["labels"].view(-1).to(DEV)
    eager = TrainStep(m1, lr=2e-3, use_graph=False)
    graph = TrainStep(m2, lr=2e-3, use_graph=True)
    le, lg = [], []
    for _ in range(12):
        le.append(eager.step(img, labels).item() / 4)
    kept = []
    for _ in range(12):
        kept.append(graph.step(img, labels))  # returned tensors must not alias the recorded one
    lg = [t.item() / 4 for t in kept]
    assert le[-1] < 0.5 * le[0], le
    # same kernels, same order, same data -> the replayed graph follows the eager trajectory from its FIRST step (the
    # capture's warm-up passes are rolled back: optimizer.snapshot / restore)
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-2 * max(abs(a), 1e-3), (le, lg)
    assert graph.optimizer.step_count == eager.optimizer.step_count == 12
    assert graph._graph is not None


def test_grad_clip_and_lazy_zero(golden):
    g, m = _small(golden)
    ts = TrainStep(m, lr=1e-3)
    img, labels = g["img"].to(DEV), g["labels"].view(-1).to(DEV)
    ts.optimizer.zero_grad()
    logits = m(img)["predictions"]
    _, dl = C.ops.softmax_xent(logits, labels, 0.25)
    logits.backward(dl)
    ts.arena.finalize_grads()
    flat = torch.cat([g["grads"][k].reshape(-1) for k, _ in m.named_parameters()])
    total = C.clip_grad_norm_(ts.arena, 1e9)
    assert abs(total.item() - flat.norm().item()) <= 2e-2 * flat.norm().item()
    # stale gradients never leak: a second lazy zero + backward gives the same arena content
    snap = ts.arena.flat_g.clone()
    ts.optimizer.zero_grad()
    logits = m(img)["predictions"]
    _, dl = C.ops.softmax_xent(logits, labels, 0.25)
    logits.backward(dl)
    ts.arena.finalize_grads()
    assert_close(ts.arena.flat_g, snap, 1e-6, "lazy zero_grad")


def test_grad_clip_is_applied_to_this_step_only(golden):
    """the clip coefficient must act on the CURRENT step (the hyper record is uploaded before backward) and
    must not compound into later steps.  Adam's first moment is linear in the gradient: after step t,
    m_t - beta1 * m_{t-1} = (1 - beta1) * clip(g_t), whose norm is (1 - beta1) * max_norm whenever |g_t| > max_norm."""
    g, m = _small(golden)
    max_norm, b1 = 0.05, 0.9
    ts = TrainStep(m, lr=1e-5, betas=(b1, 0.999), clip_norm=max_norm)
    img, labels = g["img"].to(DEV), g["labels"].view(-1).to(DEV)
    prev = torch.zeros_like(ts.optimizer.exp_avg)
    for step in range(3):
        ts.step(img, labels)
        total = ts.grad_norm.item()
        assert total > 4 * max_norm, total  # the clip is active in every step
        delta = (ts.optimizer.exp_avg - b1 * prev).norm().item() / (1 - b1)
        assert abs(delta - max_norm) <= 2e-3 * max_norm, (step, delta, max_norm)
        prev = ts.optimizer.exp_avg.clone()
    assert ts.optimizer.grad_scale == 1.0


def test_ema_eval_swaps_the_weights_the_kernels_read():
    """EMA.eval() / .train() swap `param.data` (no `_version` bump): without an arena the cached bf16
    shadows went stale and eval silently ran on the non-EMA weights."""
    from cflearn_amd.optim import EMA

    torch.manual_seed(0)
    lin = C.Linear(64, 32).to(DEV)
    x = torch.randn(16, 64, device=DEV)
    named = list(lin.named_parameters())
    ema = EMA(0.5, named).to(DEV)
    y0 = lin(x).float().clone()  # caches the bf16 shadow of the initial weights
    with torch.no_grad():
        for _, p in named:
            p.add_(0.5)  # in-place: bumps _version, the training-time path
    ema.train()
    ema()  # ema = 0.5 * old + 0.5 * new
    y_new = lin(x).float().clone()
    ema.eval()  # weights <- EMA (param.data.copy_)
    y_ema = lin(x).float().clone()
    ema.train()  # weights <- cached training weights
    y_back = lin(x).float().clone()
    assert (y_new - y0).abs().max() > 1.0
    assert_close(y_ema, 0.5 * (y0 + y_new), 2e-2, "EMA weights are the ones eval runs on")
    assert torch.equal(y_back, y_new)


def test_ema_one_kernel_bit_exact():
    """optim.EMA (reference modules/common.py:102-162): buffer names, num_updates decay rule, train/eval swap; the
    update over the arena is one kernel and bit-equal to the reference expression in fp32."""
    from cflearn_amd.optim import EMA, ParamArena

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5)).to(DEV)
    named = list(net.named_parameters())
    arena = ParamArena([p for _, p in named], with_shadow=False)
    for use_num, with_arena in ((False, True), (True, True), (False, False)):
        ema = EMA(0.999, named, use_num_updates=use_num, arena=arena if with_arena else None).to(DEV)
        assert [n for n, _ in ema.named_buffers()] == ["0_weight", "0_bias", "1_weight", "1_bias", "num_updates"]
        ref = {n.replace(".", "_"): p.detach().clone() for n, p in named}
        for step in range(3):
            with torch.no_grad():
                for _, p in named:
                    p.add_(torch.randn_like(p) * 0.1)
            ema.train()
            ema()
            decay = 0.999 if not use_num else min(0.999, (1 + step + 1) / (10 + step + 1))
            for n, p in named:
                k = n.replace(".", "_")
                ref[k] = (1.0 - decay) * p.data + decay * ref[k]
                assert torch.equal(getattr(ema, k), ref[k]), (use_num, with_arena, step, k)
        live = {n: p.detach().clone() for n, p in named}
        ema.eval()
        for n, p in named:
            assert torch.equal(p.data, ref[n.replace(".", "_")])
        ema.train()
        for n, p in named:
            assert torch.equal(p.data, live[n])


def test_train_step_with_token_count_not_multiple_of_8():
    """batch sizes whose B * T is not a multiple of 8 (37 x 17 = 629 token rows): every GEMM layout of the step —
    in particular the split-K dW form, whose K is the token count — must run and train"""
    import cflearn_amd as C
    from cflearn_amd.engine import TrainStep

    torch.manual_seed(0)
    m = C.VanillaClassifier(3, 10, 32, 128, encoder="vit", encoder_config=dict(patch_size=8, latent_dim=128, num_layers=2))
    m = m.to(DEV)
    ts = TrainStep(m, lr=1e-3)
    g = torch.Generator().manual_seed(1)
    img = torch.randn(37, 3, 32, 32, generator=g).to(DEV)
    labels = torch.randint(0, 10, (37,), generator=g).to(DEV)
    losses = [ts.step(img, labels).item() / 37 for _ in range(12)]
    assert all(l == l for l in losses) and losses[-1] < 0.7 * losses[0], losses


@pytest.mark.gpu
def test_helper_streams_have_their_own_hardware_queue():
    """functional.distinct_stream: an idle wavefront on the compute stream and on the helper stream at once takes the
    time of one (streams that share a ROCclr hardware queue would run them back to back)."""
    from cflearn_amd import functional as HF

    HF.SideStream.ensure()
    cur = torch.cuda.current_stream()
    lanes = [s for s in HF.SideStream.streams if s is not None]
    assert len(lanes) == HF.SideStream.lanes
    for s in lanes:
        assert HF._overlap([cur, s])
    assert HF._overlap([cur] + lanes)
    assert not HF._overlap([cur, cur])  # the check itself: one stream serialises


@pytest.mark.gpu
def test_optimizer_scalars_survive_a_host_that_runs_ahead(golden):
    """The Adam step reads its step-dependent scalars from a device record filled by an async copy out of pinned host
    memory.  With the GPU kept busy (an idle wavefront for 100 ms) the host enqueues many steps before the first copy
    executes: every step must still see ITS OWN bias corrections (a single host buffer would be overwritten)."""
    from cflearn_amd import ops

    def run(stall):
        g, m = _small(golden)
        ts = TrainStep(m, lr=1e-3, weight_decay=0.01)
        x, y = g["img"].to(DEV), g["labels"].view(-1).to(DEV)
        torch.cuda.synchronize()
        if stall:
            ops.spin(100000)
        for _ in range(6):
            ts.step(x, y)
            if not stall:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return ts.arena.flat_p.clone()

    a, b = run(False), run(True)
    # (LayerNorm parameter gradients use LDS float atomics: equal up to the last bits, not bitwise)
    assert_close(b, a, 2e-6, "parameters after 6 steps, stalled GPU vs step-by-step")


@pytest.mark.parametrize("width,layers,grouped", [(256, 3, False), (512, 4, True)])
def test_stack_launch_plans_replay_bit_identically(width, layers, grouped):
    """fused.StackPlan (round 4): the second step of a block stack is recorded at the C-ABI boundary, later steps replay the
    recorded launches.  Same kernels, same arguments: the trajectory of a 256-wide ViT (LayerNorm gradients on the
    deterministic kernel, batch large enough for the two batch slices, optimizer inside backward with both shadow arenas)
    must be BIT-identical with plans on and off — parameters and moments after every step, losses — and the plan must really
    have replayed.  The 512-wide case (round 5) has more than DW_MIN_TILES weight-gradient tiles per flush, so
    the stack's weight gradients go out as GROUPED launches whose operand addresses sit in a host problem table: the last
    block's dW2 reads the stack's incoming gradient through that table, and a replay that moves the incoming gradient must
    patch the table too (it did not: w2.grad / b2.grad of the last block came from the recorded step's gradient)."""
    from cflearn_amd import fused

    def run(plans: bool, steps: int = 7):
        prev = fused.STACK_PLANS
        fused.STACK_PLANS = plans
        fused._plans.clear()
        try:
            torch.manual_seed(3)
            m = C.vit_b16_classifier(10, img_size=32, patch_size=8, latent_dim=width, num_layers=layers).to(DEV)
            ts = TrainStep(m, lr=1e-3, weight_decay=0.01)
            gen = torch.Generator().manual_seed(5)
            batches = [(torch.randn(8, 3, 32, 32, generator=gen).to(DEV), torch.randint(0, 10, (8,), generator=gen).to(DEV)) for _ in range(3)]
            out = []
            for i in range(steps):
                img, lab = batches[i % 3]
                loss = ts.step(img, lab)
                torch.cuda.synchronize()
                out.append((loss.item(), ts.arena.flat_p.clone(), ts.optimizer.exp_avg_sq.clone()))
            plan = next(iter(fused._plans.values())) if fused._plans else None
            return out, plan
        finally:
            fused.STACK_PLANS = prev
            fused._plans.clear()

    ref, no_plan = run(False)
    got, plan = run(True)
    assert no_plan is None
    assert plan is not None and plan.ready_fwd and plan.ready_bwd and not plan.disabled and plan.calls == 7
    assert plan.fwd_alt is not None and len(plan.fwd) > 30 and len(plan.bwd) > 60  # both shadow arenas, real launch lists
    tables = [e for e in plan.bwd if e[0] == 3]
    assert bool(tables) == grouped, (len(tables), grouped)
    if grouped:  # the incoming gradient is an operand of a grouped launch, and the plan knows where
        assert any(type(site[2]) is tuple for site in plan.dy_sites), plan.dy_sites
    for i, ((l0, p0, v0), (l1, p1, v1)) in enumerate(zip(ref, got)):
        assert abs(l0 - l1) <= 1e-5 * abs(l0), (i, l0, l1)  # (the scalar loss is a sum of per-sample f32 atomics: not bitwise)
        assert torch.equal(p0, p1) and torch.equal(v0, v1), i


def test_stack_launch_plans_fall_back():
    """what a plan cannot follow takes the normal path: another batch size re-records, a forward-only call between steps
    (two forwards before a backward) drops the plan in flight and starts a new one — switched off for good only when it keeps
    happening —, evaluation runs without one."""
    from cflearn_amd import fused

    fused._plans.clear()
    fused._plan_conflicts.clear()
    torch.manual_seed(3)
    m = C.vit_b16_classifier(10, img_size=32, patch_size=8, latent_dim=256, num_layers=2).to(DEV)
    ts = TrainStep(m, lr=1e-3)
    gen = torch.Generator().manual_seed(5)
    img, lab = torch.randn(8, 3, 32, 32, generator=gen).to(DEV), torch.randint(0, 10, (8,), generator=gen).to(DEV)
    for _ in range(4):
        ts.step(img, lab)
    plan = next(iter(fused._plans.values()))
    assert plan.ready_bwd
    l8 = ts.step(img, lab).item()
    ts.step(img[:6], lab[:6])  # another shape: a new plan object for this stack
    plan2 = next(iter(fused._plans.values()))
    assert plan2 is not plan and not plan2.ready_fwd
    for _ in range(3):
        ts.step(img, lab)
    plan3 = next(iter(fused._plans.values()))
    assert plan3.ready_bwd
    ts.optimizer.zero_grad()
    m(img)  # forward in training mode without a backward ...
    ts.step(img, lab)  # ... then a step: the stack is asked again while the plan is in flight
    # (round 5) one abandoned forward does not cost the stack its plans: the plan in flight is dropped, a new one starts
    plan4 = next(iter(fused._plans.values()))
    assert plan4 is not plan3 and not plan4.disabled and not plan3.disabled and not plan4.ready_fwd
    for _ in range(3):
        l_after = ts.step(img, lab).item()
    assert plan4.ready_bwd and plan4.calls >= 4  # recorded again and replaying
    assert l_after == l_after and l_after < l8 * 1.5
    # a stack that keeps being called twice per backward ends on the normal path, with one warning
    import warnings as _w

    with _w.catch_warnings(record=True) as seen:
        _w.simplefilter("always")
        for _ in range(6):
            ts.optimizer.zero_grad()
            m(img)
            ts.step(img, lab)
    last = next(iter(fused._plans.values()))
    assert last.disabled and sum("launch plans are off" in str(w.message) for w in seen) == 1
    l_end = ts.step(img, lab).item()
    assert l_end == l_end and l_end < l8 * 1.5
    m.eval()
    with torch.no_grad():
        y = m(img)["predictions"]
    assert torch.isfinite(y.float()).all()
    fused._plans.clear()


def test_stack_launch_plans_follow_replaced_gradient_buffers():
    """the recorded launches carry the ADDRESSES of parameters and gradients: when somebody replaces a `.grad` (here: every
    gradient of the stack gets a new zero tensor outside the arena) the plan key changes (address fingerprint) and the
    stack records again instead of replaying into the old buffers; results equal the no-plan path bit for bit."""
    from cflearn_amd import fused

    def run(plans: bool):
        prev = fused.STACK_PLANS
        fused.STACK_PLANS = plans
        fused._plans.clear()
        try:
            torch.manual_seed(4)
            m = C.vit_b16_classifier(10, img_size=32, patch_size=8, latent_dim=256, num_layers=2).to(DEV)
            gen = torch.Generator().manual_seed(6)
            img, lab = torch.randn(8, 3, 32, 32, generator=gen).to(DEV), torch.randint(0, 10, (8,), generator=gen).to(DEV)
            grads = []
            for step in range(8):
                if step == 5:  # new homes for the gradients (a user's own accumulation buffers)
                    for p_ in m.parameters():
                        p_.grad = torch.zeros_like(p_)
                else:
                    for p_ in m.parameters():
                        if p_.grad is not None:
                            p_.grad.zero_()
                out = m(img)["predictions"]
                loss_sum, dlogits = ops.softmax_xent(out, lab, 1.0 / 8)
                out.backward(dlogits)
                HF.SideStream.join()
                torch.cuda.synchronize()
                grads.append(torch.cat([p_.grad.flatten() for p_ in m.parameters()]).clone())
            plan = next(iter(fused._plans.values())) if fused._plans else None
            return grads, plan
        finally:
            fused.STACK_PLANS = prev
            fused._plans.clear()

    ref, _ = run(False)
    got, plan = run(True)
    assert plan is not None and plan.ready_bwd  # recorded again after the switch and replayed since
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), i
