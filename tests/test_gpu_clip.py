"""CLIP towers on the GPU (quick-GELU GEMM epilogues, causal attention flag, embedding gather + positional add,
EOT row gather, L2 normalisation) vs the reference-made fixture tests/golden/clip_small.pt."""
import pytest
import torch

import cflearn_amd as C
from cflearn_amd import functional as HF
from cflearn_amd import ops
from helpers import assert_close, bf16_round, rel_l2

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def test_quick_gelu_epilogues_and_standalone():
    torch.manual_seed(0)
    m, n, k = 300, 256, 128
    a = bf16_round(torch.randn(m, k) * 0.5)
    w = bf16_round(torch.randn(n, k) * 0.1)
    b = torch.randn(n) * 0.2
    pre_want = a @ w.t() + b
    pre = torch.empty(m, n, dtype=torch.bfloat16, device=DEV)
    y = ops.gemm(a.to(DEV).bfloat16(), w.to(DEV).bfloat16(), bias=b.to(DEV), epilogue=ops.EPI_QGELU, aux_out=pre)
    assert_close(pre, pre_want, 4e-3, "pre")
    pr = pre.float().cpu()
    assert_close(y, pr * torch.sigmoid(1.702 * pr), 4e-3, "quick gelu epilogue")
    assert_close(ops.quick_gelu_fwd(pre), pr * torch.sigmoid(1.702 * pr), 4e-3, "quick gelu standalone")
    # derivative epilogue: dX = (dY W) * qgelu'(pre)
    dy = bf16_round(torch.randn(m, k) * 0.3)
    w2 = bf16_round(torch.randn(k, n) * 0.1)  # [K=k][N=n] row-major (b_trans)
    s = torch.sigmoid(1.702 * pr)
    want = (dy @ w2) * (s * (1 + 1.702 * pr * (1 - s)))
    got = ops.gemm(dy.to(DEV).bfloat16(), w2.to(DEV).bfloat16(), b_trans=True, epilogue=ops.EPI_DQGELU, aux_in=pre)
    assert_close(got, want, 5e-3, "quick gelu' epilogue")
    g1 = bf16_round(torch.randn(m, n))
    assert_close(ops.quick_gelu_bwd(g1.to(DEV).bfloat16(), pre), g1 * (s * (1 + 1.702 * pr * (1 - s))), 5e-3, "qgelu bwd")


def test_embedding_gather_scatter_l2norm():
    torch.manual_seed(1)
    v, d, b, t = 50, 64, 3, 7
    table = torch.randn(v, d)
    pos = torch.randn(1, 9, d)
    idx = torch.randint(0, v, (b, t))
    idx[0, 0] = idx[1, 3] = idx[2, 6] = 0  # padding id
    tw = table.to(DEV).requires_grad_(True)
    pw = pos.to(DEV).requires_grad_(True)
    out = HF.embedding(idx.to(DEV), tw, pw, 0)
    assert torch.equal(out.cpu(), table[idx] + pos[:, :t])  # gather + one fp32 add: exact
    gy = torch.randn(b, t, d)
    out.backward(gy.to(DEV))
    want = torch.zeros(v, d).index_add_(0, idx.reshape(-1), gy.reshape(-1, d))
    want[0] = 0
    assert_close(tw.grad, want, 1e-6, "embedding grad")
    assert_close(pw.grad[:, :t], gy.sum(0, keepdim=True), 8e-3, "pos grad")  # (summed from the bf16 stream)
    assert (pw.grad[:, t:] == 0).all()
    # row gather (EOT pooling) and its scatter
    x = torch.randn(b, t, d).to(DEV).requires_grad_(True)
    sel = torch.tensor([2, 6, 0])
    got = HF.gather_rows(x, sel.to(DEV))
    assert torch.equal(got.cpu(), x.detach().cpu()[torch.arange(b), sel])
    g2 = torch.randn(b, d)
    got.backward(g2.to(DEV))
    ref = torch.zeros(b, t, d)
    ref[torch.arange(b), sel] = g2
    assert torch.equal(x.grad.cpu(), ref)
    # L2 normalisation
    z = torch.randn(5, 96, requires_grad=True)
    zr = z / z.norm(dim=-1, keepdim=True)
    gz = torch.randn(5, 96)
    zr.backward(gz)
    zg = z.detach().to(DEV).requires_grad_(True)
    yy = HF.l2_normalize(zg)
    yy.backward(gz.to(DEV))
    assert_close(yy, zr, 1e-6, "l2norm")
    assert_close(zg.grad, z.grad, 1e-5, "l2norm grad")


def _clip(g):
    m = C.build_module("clip", config=dict(g["cfg"]))
    m.load_state_dict(g["sd"])
    return m.to(DEV)


def test_clip_towers_golden(golden):
    import clip_oracle as CL

    g = golden("clip_small.pt")
    m = _clip(g)
    assert list(m.state_dict().keys()) == list(g["sd"].keys())
    img, txt = g["img"].to(DEV), g["txt"].to(DEV)
    fi = m.encode_image(img)
    ft = m.encode_text(txt)
    assert fi.dtype == torch.float32 and ft.dtype == torch.float32
    assert_close(fi, g["image_features"], 2e-2, "image features")
    assert_close(ft, g["text_features"], 2e-2, "text features")
    logits = m(img, txt)
    assert_close(logits, g["logits"], 3e-2, "logits")
    target = torch.arange(4, device=DEV)
    loss = 0.5 * (torch.nn.functional.cross_entropy(logits, target) + torch.nn.functional.cross_entropy(logits.t(), target))
    assert abs(loss.item() - g["loss"].item()) <= 2e-2 * abs(g["loss"].item())
    loss.backward()
    grads = {k: p.grad.detach().float().cpu() for k, p in m.named_parameters() if p.grad is not None}
    assert set(grads) == set(g["grads"])
    worst = 0.0
    for k, v in grads.items():
        worst = max(worst, assert_close(v, g["grads"][k], 1.2e-1, f"grad {k}", abs_floor=3e-4))
    print(f"clip worst grad rel-L2 vs fp32 reference {worst:.3e}")
    # un-pooled text output (apply_pooling=False): the LayerNorm-ed sequence
    seq = m.encode_text(txt, apply_pooling=False)
    assert seq.shape == (4, 16, 128)


def test_clip_fused_equals_composed(golden):
    """fused stack (one autograd node, causal flag) == per-op composed path (explicit triu mask through the
    reference's mask expansion): same kernels underneath, so agreement is tight"""
    g = golden("clip_small.pt")
    outs = []
    for fused in (True, False):
        m = _clip(g)
        for enc in (m.vit.encoder, m.text_transformer.encoder):
            for blk in enc.mixing_blocks:
                blk.use_fused = fused
        logits = m(g["img"].to(DEV), g["txt"].to(DEV))
        logits.sum().backward()
        outs.append((logits.detach(), {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()
                                       if p.grad is not None}))
    assert_close(outs[0][0], outs[1][0], 5e-3, "fused vs composed logits")
    for k in outs[0][1]:
        assert_close(outs[0][1][k], outs[1][1][k], 3e-2, f"fused vs composed {k}", abs_floor=2e-4)


def test_contrastive_loss_kernels_vs_fp32_formula():
    """contrastive.ClipLossFn / SimilarityFn (fp32 similarity GEMMs + softmax-CE + dot) against the plain fp32
    statement of the symmetric InfoNCE (oracle/clip_oracle.contrastive_loss_local), ragged sizes."""
    import clip_oracle as CL
    from cflearn_amd.contrastive import clip_contrastive_loss, similarity_logits

    for b, d in ((19, 40), (32, 64), (5, 7)):
        gen = torch.Generator().manual_seed(b * 7 + d)
        img = torch.nn.functional.normalize(torch.randn(b, d, generator=gen), dim=-1)
        txt = torch.nn.functional.normalize(torch.randn(b, d, generator=gen), dim=-1)
        ir, tr = img.clone().requires_grad_(True), txt.clone().requires_grad_(True)
        lsr = torch.tensor(2.0, requires_grad=True)
        want = CL.contrastive_loss_local(ir, tr, ir, tr, lsr)
        want.backward()
        idv, tdv = img.to(DEV).requires_grad_(True), txt.to(DEV).requires_grad_(True)
        lsd = torch.tensor(2.0, device=DEV, requires_grad=True)
        got = clip_contrastive_loss(idv, tdv, lsd)
        assert got.shape == (1,) and got.dtype == torch.float32
        assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item())
        (got * 3.0).sum().backward()  # a non-trivial upstream gradient
        assert_close(idv.grad, 3.0 * ir.grad, 1e-5, "d image features")
        assert_close(tdv.grad, 3.0 * tr.grad, 1e-5, "d text features")
        assert abs(lsd.grad.item() - 3.0 * lsr.grad.item()) <= 1e-4 * abs(3.0 * lsr.grad.item()) + 1e-6
        # the logits alone, with their own backward
        i2, t2 = img.to(DEV).requires_grad_(True), txt.to(DEV).requires_grad_(True)
        ls2 = torch.tensor(2.0, device=DEV, requires_grad=True)
        lg = similarity_logits(i2, t2, ls2)
        ref = (lsr.detach().exp() * img @ txt.t())
        assert_close(lg, ref, 1e-6, "similarity logits")
        w = torch.randn(b, b, generator=gen)
        (lg * w.to(DEV)).sum().backward()
        ir2, tr2, ls3 = img.clone().requires_grad_(True), txt.clone().requires_grad_(True), torch.tensor(2.0, requires_grad=True)
        ((ls3.exp() * ir2 @ tr2.t()) * w).sum().backward()
        assert_close(i2.grad, ir2.grad, 1e-5, "similarity d image")
        assert_close(t2.grad, tr2.grad, 1e-5, "similarity d text")
        assert abs(ls2.grad.item() - ls3.grad.item()) <= 1e-4 * abs(ls3.grad.item()) + 1e-6


def test_clip_contrastive_step_golden(golden):
    """CLIP.contrastive_loss (single process) against the fixture's loss and parameter gradients — the fixture was
    produced with torch's cross_entropy on the reference's logits_per_image and its transpose."""
    g = golden("clip_small.pt")
    m = _clip(g)
    loss = m.contrastive_loss(g["img"].to(DEV), g["txt"].to(DEV))
    assert abs(loss.item() - g["loss"].item()) <= 2e-2 * abs(g["loss"].item())
    loss.sum().backward()
    grads = {k: p.grad.detach().float().cpu() for k, p in m.named_parameters() if p.grad is not None}
    assert set(grads) == set(g["grads"])
    for k, v in grads.items():
        assert_close(v, g["grads"][k], 1.2e-1, f"grad {k}", abs_floor=3e-4)


def test_clip_contrastive_train_steps(golden):
    """engine.LossTrainStep on CLIP.contrastive_loss: arena + fused AdamW, first loss = the fixture's, then it falls."""
    from cflearn_amd.engine import LossTrainStep

    g = golden("clip_small.pt")
    m = _clip(g)
    ts = LossTrainStep(m, lambda mod, b: mod.contrastive_loss(b["image"], b["text"]), lr=1e-3)
    batch = dict(image=g["img"].to(DEV), text=g["txt"].to(DEV))
    losses = [ts.step(batch).item() for _ in range(6)]
    assert abs(losses[0] - g["loss"].item()) <= 2e-2 * abs(g["loss"].item())
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    # logit_scale (a 0-d parameter whose gradient comes back through autograd) lives in the arena and moved
    assert m.logit_scale.grad is not None and m.logit_scale.grad.data_ptr() >= ts.arena.flat_g.data_ptr()
    assert abs(m.logit_scale.item() - g["sd"]["logit_scale"].item()) > 1e-4


def test_clip_step_updates_inside_backward_like_the_end_of_step_launch(golden):
    """LossTrainStep(step_in_backward=True) (round 4): arena ranges are updated inside backward as their last gradient is
    announced; the parameters after three steps equal the end-of-step launch's.  CLIP's embedding-table gradients are f32
    atomic scatter-adds (run-to-run differences in the last bits), hence a tolerance instead of torch.equal; `logit_scale`
    (gradient through autograd's accumulation: never announced) is covered by the complement launch."""
    from cflearn_amd.engine import LossTrainStep

    g = golden("clip_small.pt")
    batch = dict(image=g["img"].to(DEV), text=g["txt"].to(DEV))
    outs = []
    for in_bwd in (False, True):
        m = _clip(g)
        ts = LossTrainStep(m, lambda mod, b: mod.contrastive_loss(b["image"], b["text"]), lr=1e-3, step_in_backward=in_bwd,
                           range_bytes=1 << 20)
        losses = [ts.step(batch).item() for _ in range(3)]
        if in_bwd:
            assert ts.optimizer.in_backward.launched_in_backward > 0
        torch.cuda.synchronize()
        outs.append((losses, ts.arena.flat_p.clone()))
    (l0, p0), (l1, p1) = outs
    assert max(abs(a - b) for a, b in zip(l0, l1)) <= 2e-3 * abs(l0[0]), (l0, l1)
    assert_close(p1, p0, 2e-4, "parameters after three steps")


# ---------------------------------------------------------------------------------------------------------------------
# Full size (round 6): the model bench.py --workload clip times, against the fp32 oracle on the host cores and
# against the REFERENCE'S OWN bf16-autocast distance from fp32 (tests/golden/clip_b32_yardstick.pt, made in the build container
# from cflearn's CLIP by oracle/gen_clip_b32_yardstick.py).
# ---------------------------------------------------------------------------------------------------------------------
_B32 = {}


def _clip_b32_oracle():
    """seeded problem + fp32 oracle (features, logits, loss, sampled gradients) on the host cores, once per session"""
    if _B32:
        return _B32
    import os
    import time

    import clip_oracle as CL
    from gen_clip_b32_yardstick import BATCH, CLIP_SAMPLED, info_nce, probe, seeded_problem

    ref = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_b32_yardstick.pt"), weights_only=False)
    assert ref["batch"] == BATCH and list(ref["grad_err"]) == CLIP_SAMPLED
    sd, img, txt = seeded_problem()
    assert ref["n_params"] == 151277825  # ViT-B/32 + 12 x 512 text tower
    prev = torch.get_num_threads()
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    try:
        t0 = time.time()
        osd = {k: v.clone() for k, v in sd.items()}
        leaves = [osd[k].requires_grad_(True) for k in CLIP_SAMPLED]
        fi = CL.encode_image(img, osd, 12, 12)
        ft = CL.encode_text(txt, osd, 8, 12)
        logits = osd["logit_scale"].exp() * fi @ ft.t()
        loss = info_nce(logits)
        grads = torch.autograd.grad(loss, leaves)
        print(f"CLIP ViT-B/32 + text oracle, batch {BATCH}: fp32 forward + backward on the host {time.time() - t0:.1f} s")
    finally:
        torch.set_num_threads(prev)
    # (1) the restatement against the reference's own fp32 run at FULL size (two fp32 CPU runs with different op decompositions)
    assert abs(loss.item() - ref["loss_fp32"]) <= 1e-5 * abs(ref["loss_fp32"]), (loss.item(), ref["loss_fp32"])
    assert rel_l2(probe(fi), ref["image_features_probe"]) <= 1e-4
    assert rel_l2(probe(ft), ref["text_features_probe"]) <= 1e-4
    assert rel_l2(probe(logits), ref["logits_probe"]) <= 1e-4
    for k, gr in zip(CLIP_SAMPLED, grads):
        assert rel_l2(probe(gr), ref["grad_probe"][k]) <= 5e-4, (k, rel_l2(probe(gr), ref["grad_probe"][k]))
        assert abs(gr.norm().item() - ref["grad_norm"][k]) <= 2e-4 * ref["grad_norm"][k], k
    _B32.update(ref=ref, sd=sd, img=img, txt=txt, fi=fi.detach(), ft=ft.detach(), logits=logits.detach(), loss=loss.item(),
                grads=dict(zip(CLIP_SAMPLED, (g_.detach() for g_ in grads))), names=CLIP_SAMPLED)
    return _B32


@pytest.mark.parametrize("towers", [True, False], ids=["towers_side_by_side", "one_tower_after_the_other"])
def test_clip_b32_step_vs_oracle(towers):
    """BASELINE config 5 at the benchmarked model size (multimodal/clip.py:209-256, nlp/encoder/transformer.py:17-99: ViT-B/32 at
    224^2 = 50 tokens x 768 + 12 x 512 causal text tower at 77 tokens, vocabulary 49 408, ragged captions with the end token
    placed as SURVEY §8(d) says), batch 16, through the path bench.py times: `LossTrainStep` over `CLIP.contrastive_loss` ->
    `_encode_both` (arena + bf16 shadows, in-backward optimizer, the recorded AND the replayed launch plan), with the two towers
    side by side on two streams and one after the other (`CFHIP_CLIP_TOWERS=0`).  lr = 0 keeps the problem fixed over the three
    steps, so step 1 (composed launch), step 2 (plan recording) and step 3 (plan replay) must all reproduce the oracle."""
    from cflearn_amd.engine import LossTrainStep

    o = _clip_b32_oracle()
    ref, names = o["ref"], o["names"]
    torch.manual_seed(0)
    m = C.build_module("clip", config={})
    m.load_state_dict(o["sd"])
    m = m.to(DEV)
    m.towers_side_by_side = towers
    img, txt = o["img"].to(DEV), o["txt"].to(DEV)
    with torch.no_grad():
        fi, ft = m.encode_image(img), m.encode_text(txt)
        logits = m(img, txt)
    e_fi, e_ft, e_lg = rel_l2(fi, o["fi"]), rel_l2(ft, o["ft"]), rel_l2(logits, o["logits"])
    print(f"CLIP b32 (towers side by side: {towers}): image features {e_fi:.3e} (reference bf16-autocast {ref['image_features_err']:.3e}), "
          f"text features {e_ft:.3e} ({ref['text_features_err']:.3e}), logits {e_lg:.3e} ({ref['logits_err']:.3e}, its quadrants up to {max(ref['logits_err_quadrants']):.3e})")
    assert e_fi <= 1.1 * ref["image_features_err"], (e_fi, ref["image_features_err"])
    assert e_ft <= 1.1 * ref["text_features_err"], (e_ft, ref["text_features_err"])
    # (the logits of 16 near-identical image features against 16 captions are ~16 numbers: the reference's own quadrants spread
    # 0.77 .. 1.15 x around its whole-matrix distance — oracle/gen_clip_b32_yardstick.py; bound = 1.1 x the largest of them)
    assert e_lg <= 1.1 * max(ref["logits_err_quadrants"]), (e_lg, ref["logits_err"], ref["logits_err_quadrants"])
    want_lg = (m.logit_scale.detach().exp() * fi @ ft.t()).float().cpu()
    assert rel_l2(logits, want_lg) <= 1e-5  # ... and the similarity kernel itself is exact fp32 on OUR features

    ts = LossTrainStep(m, lambda mod, b: mod.contrastive_loss(b["image"], b["text"]), lr=0.0)
    batch = dict(image=img, text=txt)
    params = dict(m.named_parameters())
    before = ts.arena.flat_p.clone()
    for step in (1, 2, 3):
        loss = ts.step(batch).item()
        torch.cuda.synchronize()
        loss_err = abs(loss - o["loss"]) / abs(o["loss"])
        errs = {k: rel_l2(params[k].grad, o["grads"][k]) for k in names}
        worst = max(errs, key=lambda k: errs[k] / ref["grad_err"][k])
        print(f"  step {step}: loss {loss:.6f} vs {o['loss']:.6f} (rel {loss_err:.2e}); worst sampled gradient {errs[worst]:.3e} = "
              f"{errs[worst] / ref['grad_err'][worst]:.2f} x the reference's bf16-autocast distance ({worst})")
        if step in (1, 3):
            for k in names:
                print(f"    {k:82s} {errs[k]:.3e}   reference bf16-autocast {ref['grad_err'][k]:.3e} (x {errs[k] / ref['grad_err'][k]:.2f})")
        assert loss_err <= 1e-3, (loss, o["loss"])
        for k in names:
            assert errs[k] <= max(1e-2, 1.1 * ref["grad_err"][k]), (step, k, errs[k], ref["grad_err"][k])
    assert torch.equal(ts.arena.flat_p, before)  # lr = 0: the three steps really were the same problem
    assert ts.optimizer.in_backward is None or ts.optimizer.in_backward.launched_in_backward > 0
