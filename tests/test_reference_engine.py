"""The reference's OWN package (imported from /root/reference through oracle/refharness) on CPU:

  * BASELINE config 1: `cflearn.api.fit_ml` FCNN on synthetic tabular data (api/api.py:496-526), CPU, world size 1;
  * the drop-in boundary: INTEGRATION.md §2's registry override executed on the reference's real `module_dict`, built
    through the reference's `build_module` (modules/common.py:37-53);
  * the pin of `oracle/trainer_oracle.py` (the step-engine restatement the GPU tests drive the HIP modules with)
    against the reference's `IDLModel.train` / `get_update_fn` / `Trainer.clip_norm_step` (schema.py:977-986,
    1174-1294; trainer.py:170-176): bit-equal losses and weights on the same model and batches;
  * the reference's gradient checkpointing (`toolkit.py:2535-2647`) around modules of this package (row U6).

Build container only (`/root/reference` is absent on the GPU box): every test here is `not gpu`.
"""
import os
import sys
import types
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import refharness  # noqa: E402
import trainer_oracle as TO  # noqa: E402

pytestmark = pytest.mark.skipif(not refharness.reference_available(), reason="/root/reference is not present")


@pytest.fixture(scope="module")
def cflearn():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return refharness.import_cflearn()


def test_reference_fit_ml_fcnn_cpu(cflearn, tmp_path, monkeypatch):
    """BASELINE.json configs[0]: FCNN tabular classifier on synthetic data via cflearn.api (SURVEY §8d inputs)."""
    monkeypatch.chdir(tmp_path)
    x = np.random.RandomState(0).randn(2000, 32).astype(np.float32)
    w = np.random.RandomState(1).randn(32, 3)
    y = (x @ w).argmax(1).reshape(-1, 1)  # learnable labels: the loss must go down
    config = cflearn.MLConfig(module_name="fcnn", module_config=dict(input_dim=32, output_dim=3), loss_name="focal",
                              fixed_steps=200, tqdm_settings=None)
    m = cflearn.api.fit_ml(x, y, config=config, device="cpu")
    loader = m.data.build_loader(x, y)
    preds = m.predict(loader)[TO.PREDICTIONS_KEY]
    assert preds.shape == (2000, 3)
    acc = (preds.argmax(1) == y.ravel()).mean()
    assert acc > 0.8, acc  # 200 steps of the reference's own trainer on the CPU learn the linear rule


def test_registry_override_on_the_reference_module_dict(cflearn):
    """INTEGRATION.md §2 on the reference's REAL registry: after the override, the reference's own `build_module`
    (the only way its models create modules) hands out this package's classes, which keep the reference's
    constructor keywords and state_dict keys."""
    import cflearn_amd as C

    common = sys.modules["cflearn.modules.common"]
    cfg = dict(input_dim=10, output_dim=3, hidden_units=[16, 16])
    ref_fcnn = common.build_module("fcnn", config=dict(cfg))
    ref_keys = list(ref_fcnn.state_dict().keys())
    saved = dict(common.module_dict)
    try:
        n = C.override_reference_registry(common.module_dict)
        assert n >= 5
        ours = common.build_module("fcnn", config=dict(cfg, some_unknown_keyword=1))  # safe_execute drops extras
        assert type(ours).__module__.startswith("cflearn_amd")
        assert list(ours.state_dict().keys()) == ref_keys
        ours.load_state_dict(ref_fcnn.state_dict())  # checkpoints move both ways
        clf = common.build_module("cv_clf", config=dict(in_channels=1, num_classes=10, img_size=28, latent_dim=64,
                                                        encoder="vanilla_1d",
                                                        encoder_config=dict(num_downsample=3)))
        assert type(clf).__module__.startswith("cflearn_amd")
    finally:
        common.module_dict.clear()
        common.module_dict.update(saved)
    assert type(common.build_module("fcnn", config=dict(cfg))).__module__.startswith("cflearn.")


def _mlp() -> torch.nn.Module:
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))


def _batches(n: int):
    g = torch.Generator().manual_seed(7)
    return [{TO.INPUT_KEY: torch.randn(8, 12, generator=g), TO.LABEL_KEY: torch.randint(0, 4, (8, 1), generator=g)}
            for _ in range(n)]


@pytest.mark.parametrize("loss_name,grad_accumulate,clip_norm", [("focal", 1, 0.0), ("cross_entropy", 2, 0.05),
                                                                  ("focal", 3, 0.1)])
def test_step_engine_oracle_matches_the_reference(cflearn, loss_name, grad_accumulate, clip_norm):
    """`oracle/trainer_oracle.StepEngine` against the reference's `CommonDLModel.train` driven by a trainer object
    that carries the reference's own `Trainer.clip_norm_step`: same losses (the `.item()` values) and bit-equal weights
    after 6 batches, with gradient accumulation and clipping."""
    CommonDLModel = sys.modules["cflearn.models.common"].CommonDLModel
    build_loss = sys.modules["cflearn.losses"].build_loss
    Trainer = sys.modules["cflearn.trainer"].Trainer

    # reference side
    ref = CommonDLModel()
    ref.m = _mlp()
    ref.loss = build_loss(loss_name)
    opt_ref = torch.optim.SGD(ref.m.parameters(), lr=0.1)

    class _Accel:
        sync_gradients = True

        @staticmethod
        def backward(loss):
            loss.backward()

        @staticmethod
        def clip_grad_norm_(params, max_norm):
            return torch.nn.utils.clip_grad_norm_(params, max_norm)

    trainer = types.SimpleNamespace(state=TO.State(), accelerator=_Accel(), optimizers={"all": opt_ref},
                                    config=TO.Config(grad_accumulate, clip_norm), should_autocast=False,
                                    model_for_training=ref.m, scheduler_step=lambda: None, _gradient_norm=None)
    trainer.clip_norm_step = types.MethodType(Trainer.clip_norm_step, trainer)  # the reference's own method
    ref_losses = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # torch.cuda.amp.autocast deprecation inside the reference
        for i, batch in enumerate(_batches(6)):
            trainer.state.step += 1
            out = ref.train(i, batch, trainer, {}, {})
            ref_losses.append(out.loss_dict[TO.LOSS_KEY])

    # restatement
    mine = _mlp()
    eng = TO.StepEngine(mine, loss_name, torch.optim.SGD(mine.parameters(), lr=0.1), grad_accumulate=grad_accumulate,
                        clip_norm=clip_norm)
    eng.fit(_batches(6))
    assert [d[TO.LOSS_KEY] for d in eng.loss_log] == ref_losses
    for a, b in zip(mine.parameters(), ref.m.parameters()):
        assert torch.equal(a, b)
    for a, b in zip(mine.parameters(), _mlp().parameters()):
        assert not torch.equal(a, b)  # and training did move them


def test_reference_gradient_checkpoint_reenters_custom_functions(cflearn):
    """Row U6: the reference's `gradient_checkpoint` (toolkit.py:2535-2647) re-runs the wrapped forward inside
    backward under `torch.autograd.grad` — a second, re-entrant trip through every custom autograd Function in it.
    The parameter-gradient protocol of this package's Functions (write straight into `.grad`, return None, run the
    gradient-ready callbacks) must survive that: exercised here with the CPU stand-in that follows the same protocol
    (tests/test_ddp_gloo._DirectLinear); the HIP Functions themselves are covered on the GPU by
    tests/test_gpu_unet.py::test_gradient_checkpoint_matches_plain_backward."""
    from test_ddp_gloo import _DirectLinear

    gradient_checkpoint = sys.modules["cflearn.toolkit"].gradient_checkpoint
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 6)
    for p in lin.parameters():
        p.grad = torch.zeros_like(p)
    x = torch.randn(5, 6, requires_grad=True)

    def block(inp):
        return torch.tanh(_DirectLinear.apply(inp, lin.weight, lin.bias))

    y = gradient_checkpoint(block, (x,), tuple(lin.parameters()), True)
    y.sum().backward()
    gw, gb, gx = lin.weight.grad.clone(), lin.bias.grad.clone(), x.grad.clone()
    # plain autograd on the same math
    lin2 = torch.nn.Linear(6, 6)
    lin2.load_state_dict(lin.state_dict())
    x2 = x.detach().clone().requires_grad_(True)
    torch.tanh(lin2(x2)).sum().backward()
    assert torch.allclose(gx, x2.grad, atol=1e-6)
    assert torch.allclose(gw, lin2.weight.grad, atol=1e-6) and torch.allclose(gb, lin2.bias.grad, atol=1e-6)


def test_lazy_loss_patch_under_the_reference_trainer(cflearn, tmp_path, monkeypatch):
    """(f)3 for cflearn users: `compat.patch_lazy_losses()` removes the per-step `.item()` of models/common.py:40-43;
    the reference's own trainer still trains, logs and snapshots (the values are read only when consumed)."""
    from cflearn_amd import compat

    monkeypatch.chdir(tmp_path)
    x = np.random.RandomState(0).randn(1000, 16).astype(np.float32)
    w = np.random.RandomState(1).randn(16, 3)
    y = (x @ w).argmax(1).reshape(-1, 1)
    seen = []
    orig = compat.patch_lazy_losses()
    try:
        common = sys.modules["cflearn.models.common"]
        wrapped = common.CommonTrainStep.loss_fn

        def spy(self, m_, state, *a, **k):
            out = wrapped(self, m_, state, *a, **k)
            if state is not None:  # a TRAINING step (evaluation passes state=None and consumes its losses at once)
                seen.append(out.losses[TO.LOSS_KEY])
            return out

        common.CommonTrainStep.loss_fn = spy
        config = cflearn.MLConfig(module_name="fcnn", module_config=dict(input_dim=16, output_dim=3), loss_name="focal",
                                  fixed_steps=60, tqdm_settings=None)
        m = cflearn.api.fit_ml(x, y, config=config, device="cpu")
    finally:
        compat.restore_losses(orig)
    assert len(seen) == 60 and all(isinstance(v, compat.LazyFloat) for v in seen)
    # per-step values are read back only where the trainer consumes them (monitor / logging steps), not every step
    assert sum(v.is_materialized for v in seen) < len(seen), [v.is_materialized for v in seen]
    preds = m.predict(m.data.build_loader(x, y))[TO.PREDICTIONS_KEY]
    assert (preds.argmax(1) == y.ravel()).mean() > 0.6
    assert float(seen[-1]) < float(seen[0])


@pytest.mark.parametrize("method", ["auto_prune", "hard_prune", "soft_prune", "simplified", "surgery"])
def test_pruner_matches_the_reference_bit_for_bit(cflearn, method):
    """`Linear(pruner_config=...)` (customs.py:54-62,84-96): the weight mask of `modules.Pruner` against the reference's
    `Pruner` (customs.py:317-413) — same state keys and initial values, bit-equal masked weight, bit-equal gradients to the
    weight and (auto_prune) to the four learnable scalars.  Pure weight-space math: torch ops on both sides."""
    import cflearn_amd as C

    ref = sys.modules["cflearn.modules.core.customs"]
    a, b = ref.Pruner({"method": method}, [12, 20]), C.modules.Pruner({"method": method}, [12, 20])
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb) or set(sa) == set(sb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    torch.manual_seed(3)
    w = torch.randn(12, 20)
    wa, wb = w.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ya, yb = a(wa), b(wb)
    assert torch.equal(ya, yb)
    g = torch.randn_like(w)
    ya.backward(g)
    yb.backward(g)
    assert torch.equal(wa.grad, wb.grad)
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    assert set(pa) == set(pb) and (set(pa) == {"alpha", "beta", "gamma", "max_ratio"}) == (method == "auto_prune")
    for k in pa:
        assert torch.equal(pa[k].grad, pb[k].grad), k
    lin_a, lin_b = ref.Linear(20, 12, pruner_config={"method": method}), C.Linear(20, 12, pruner_config={"method": method})
    assert set(lin_a.state_dict()) == set(lin_b.state_dict())


@pytest.mark.parametrize("kw", [
    dict(kernel_size=3, padding="reflection"), dict(kernel_size=3, padding="reflection2", bias=False),
    dict(kernel_size=3, transform_kernel=True), dict(kernel_size=3, padding="reflection", transform_kernel=True),
    dict(kernel_size=3, demodulate=True), dict(kernel_size=3, weight_scale=0.37),
    dict(kernel_size=3, bias=False, demodulate=True, style=True), dict(kernel_size=1, bias=False, style=True),
    dict(kernel_size=4, stride=2, padding=1, transpose=True), dict(kernel_size=3, padding=1, bias=False, transpose=True, weight_scale=2.0),
])
def test_conv2d_option_logic_matches_the_reference_with_torch_standing_in_for_the_kernels(cflearn, kw, monkeypatch):
    """The round-5 Conv2d options (convs/basic.py:41-177): everything AROUND the convolution kernel — which padding, which
    weight the convolution sees (kernel transform, style modulation, demodulation, scale), the grouped-over-the-batch reshape of the
    stylised form, the transposed weight — with the three HIP entry points replaced by their torch definitions
    (`F.pad(mode="reflect")`, `F.conv2d`, `F.conv_transpose2d`), against the reference module on the same weights and input.
    (The kernels themselves are compared with the same torch calls on the GPU: tests/test_gpu_conv.py.)"""
    import torch.nn.functional as F

    import cflearn_amd as C
    from cflearn_amd import functional as HF

    kw = dict(kw)
    style_on, transpose = kw.pop("style", False), kw.pop("transpose", False)
    refmod = sys.modules["cflearn.modules.core.convs.basic"]
    torch.manual_seed(5)
    a = refmod.Conv2d(6, 8, **kw)
    b = C.Conv2d(6, 8, **kw)
    b.load_state_dict(a.state_dict())
    monkeypatch.setattr(HF, "reflect_pad2d", lambda x, pads: F.pad(x, list(pads), mode="reflect"))
    monkeypatch.setattr(HF, "conv2d", lambda x, w, bias, s, p, d=1, g=1: F.conv2d(x, w, bias, stride=s, padding=p, dilation=d, groups=g))
    monkeypatch.setattr(HF, "conv_transpose2d", lambda x, wt, s, p, d=1: F.conv_transpose2d(x, wt, None, stride=s, padding=p, dilation=d))
    x = torch.randn(3, 6, 9, 9)
    style = torch.randn(3, 6) if style_on else None
    ya = a(x, style, transpose=transpose)
    yb = b(x, style, transpose=transpose)
    assert ya.shape == yb.shape
    assert (ya - yb).abs().max() <= 1e-5 * max(1.0, ya.abs().max().item()), (ya - yb).abs().max()
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    assert (a.weight.grad - b.weight.grad).abs().max() <= 1e-4 * a.weight.grad.abs().max()
