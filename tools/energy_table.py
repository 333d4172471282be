"""Joules per launch and per step of every kernel family of the ViT-B/16 training step — MEASUREMENT TOOL.

DESIGN §9 argues that the step runs against the socket's power cap and therefore follows the ENERGY of a step.  That premise
decided which optimisations were built in round 5 and rested on one number (1 345 W over the whole step).  This tool measures
it family by family: every launch shape of the step — the GEMM shapes with their epilogues, the grouped weight-gradient launch,
attention forward / dQ / dK,dV, LayerNorm forward / backward, the Adam launch — is looped ALONE for >= `--seconds` at steady state
with the socket power, shader clock and launch rate sampled (tools/gpu_telemetry.py, 10 Hz):

    J per launch = (W - W_idle) x seconds per launch          J per step = J per launch x launches per step

and the sum is held against the step itself: (W_step - W_idle) x seconds per step, measured the same way on the real training
step.  "Idle" = this process alive, queues empty, 3 s.  The table names the joules-per-FLOP (GEMM-class rows) and
joules-per-byte (bandwidth rows) offenders.  Caveat printed with the table: alone a kernel runs at ITS clock (often the boost
clock), in the step everything runs at ~2.0 GHz — a kernel's energy per launch is only weakly clock-dependent (same switched
capacitance, V^2 differs), which is what the coverage row checks.

    python tools/energy_table.py [--batch 128] [--seconds 2.0] [--out profiles/r06/energy_table.txt]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from tools.gpu_telemetry import GpuTelemetry  # noqa: E402


def loop(fn, seconds: float, tel: GpuTelemetry, chunk: int = 20):
    """run `fn` back to back for `seconds`; (launches, seconds per launch by HIP events, telemetry of the second half of the window:
    the first half is the power / clock ramp)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(chunk):
            fn()
        n += chunk
        torch.cuda.synchronize()
    e1.record()
    e1.synchronize()
    t1 = time.perf_counter()
    return n, e0.elapsed_time(e1) * 1e-3 / n, tel.summary(t0 + 0.5 * (t1 - t0), t1)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--step-seconds", type=float, default=6.0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import cflearn_amd as C
    from cflearn_amd import fused, ops
    from cflearn_amd.engine import TrainStep

    dev = torch.device("cuda")
    tel = GpuTelemetry(0, hz=10.0).start()
    if not tel.available:
        raise SystemExit(f"no power telemetry on this box: {tel.error}")
    lines = []

    def out(s: str = "") -> None:
        print(s, flush=True)
        lines.append(s)

    time.sleep(1.0)
    t0 = time.perf_counter()
    time.sleep(3.0)
    idle = tel.summary(t0, time.perf_counter())
    w_idle = idle["power_w_avg"]
    out(f"# energy table, ViT-B/16 224^2 batch {args.batch}, one MI355X; power cap {idle['power_cap_w']} W; idle (process alive, queues empty, 3 s): "
        f"{w_idle} W at {idle['sclk_mhz_avg']} MHz; source {idle['source']}")

    # ---- the step itself
    B, T, D, H, L = args.batch, 197, 768, 12, 12
    torch.manual_seed(0)
    model = C.vit_b16_classifier(1000).to(dev)
    ts = TrainStep(model, lr=1.0e-4)
    g = torch.Generator().manual_seed(1234)
    ring = [(torch.randn(B, 3, 224, 224, generator=g).to(dev), torch.randint(0, 1000, (B,), generator=g).to(dev)) for _ in range(4)]
    i = [0]

    def step() -> None:
        i[0] += 1
        ts.step(*ring[i[0] % 4])

    n, sec, st = loop(step, args.step_seconds, tel, chunk=5)
    j_step = (st["power_w_avg"] - w_idle) * sec
    out(f"# the training step: {sec * 1e3:.3f} ms / step over {n} steps, {st['power_w_avg']} W at {st['sclk_mhz_avg']} MHz (range {st['sclk_mhz_range']}), "
        f"junction {st['junction_c_avg']} C  =>  {st['power_w_avg'] * sec:.2f} J / step total, {j_step:.2f} J / step above idle")
    words = fused.GRAD_STREAM_WORDS
    del ts, model
    fused._plans.clear()
    torch.cuda.empty_cache()

    rows = []  # (name, class, count per step, sec per launch, W, sclk, J per launch, flops per launch, bytes per launch)

    def measure(name: str, cls: str, count: float, fn, flops: float, nbytes: float) -> None:
        time.sleep(0.3)
        n_, sec_, s_ = loop(fn, args.seconds, tel)
        w = s_["power_w_avg"] or 0.0
        rows.append(dict(name=name, cls=cls, count=count, us=sec_ * 1e6, w=w, sclk=s_["sclk_mhz_avg"], j=(w - w_idle) * sec_, flops=flops, bytes=nbytes))
        r = rows[-1]
        out(f"  {name:64s} x{count:5.0f}  {r['us']:8.1f} us  {w:7.1f} W  {r['sclk'] or 0:6.0f} MHz  {r['j'] * 1e3:8.2f} mJ/launch  {r['j'] * count:7.3f} J/step"
            + (f"  {r['j'] / flops * 1e12:6.3f} J/TFLOP" if flops else f"  {r['j'] / nbytes * 1e9:6.3f} J/GB"))

    out(f"# kernels alone, {args.seconds} s each (telemetry of the second half of each window)")
    out(f"  {'launch':64s} {'count':>6s}  {'time':>11s}  {'power':>9s}  {'sclk':>10s}  {'energy':>17s}  {'per step':>14s}  per unit")
    for desc, fn, count, flops, nbytes in bench.gemm_launchers(B):
        if desc["layout"] == "tn-grouped":
            name = f"grouped dW, {len(desc['problems'])} problems (K = {desc['problems'][0][2]})"
        else:
            name = f"gemm {desc['layout']} {desc['M']}x{desc['N']}x{desc['K']} {desc['epilogue']}"
        measure(name, "gemm", count, fn, flops, nbytes)
    # attention: one launch per batch slice and layer
    nsl = fused.BWD_HALVES
    bs = B // nsl
    g2 = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(bs, T, 3 * D, device=dev, generator=g2).to(torch.bfloat16)
    d_o = torch.randn(bs, T, D, device=dev, generator=g2).to(torch.bfloat16)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    o, lse = ops.attn_fwd(q, k, v, H)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
    delta = torch.empty(bs, H, T, device=dev)
    af = 4.0 * bs * H * T * T * 64
    measure(f"attention forward, {bs} x {H} heads x {T}", "attn", L * nsl, lambda: ops.attn_fwd(q, k, v, H), af, 0)
    measure("attention backward dQ", "attn", L * nsl, lambda: ops.attn_bwd(q, k, v, o, d_o, lse, H, dq=dq, dk=dk, dv=dv, parts=1, delta=delta), 1.5 * af, 0)
    measure("attention backward dK, dV", "attn", L * nsl, lambda: ops.attn_bwd(q, k, v, o, d_o, lse, H, dq=dq, dk=dk, dv=dv, parts=2, delta=delta), 2.0 * af, 0)
    # LayerNorm: f32 stream rows of one slice
    m = bs * T
    x = torch.randn(m, D, device=dev, generator=g2)
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    y, mean, rstd = ops.layernorm_fwd(x, gam, bet, 1e-6)
    measure(f"LayerNorm forward {m} x {D} (f32 rows -> bf16)", "hbm", 2 * L * nsl, lambda: ops.layernorm_fwd(x, gam, bet, 1e-6, out=y, mean=mean, rstd=rstd), 0, m * D * 6.0)
    dy = torch.randn(m, D, device=dev, generator=g2).to(torch.bfloat16)
    add = torch.randn(m, D, device=dev, generator=g2)
    ahi, alo = ops.split_f32(add)
    dxh, dxl = torch.empty_like(ahi), torch.empty_like(ahi)
    pg = torch.zeros(2 * D, device=dev)
    measure(f"LayerNorm backward {m} x {D}, one-word gradient stream", "hbm", 2 * L * nsl,
            lambda: ops.layernorm_bwd(dy, x, gam, mean, rstd, dx_add=ahi, dx_out=dxh, dgamma=pg[:D], dbeta=pg[D:], accumulate=True), 0, m * D * 10.0)
    measure(f"LayerNorm backward {m} x {D}, two-word gradient stream", "hbm", 2 * L * nsl,
            lambda: ops.layernorm_bwd(dy, x, gam, mean, rstd, dx_add=ahi, dx_add_lo=alo, dx_out=dxh, dx_lo_out=dxl, dgamma=pg[:D], dbeta=pg[D:],
                                      accumulate=True), 0, m * D * 14.0)
    # Adam over the whole arena (10 range launches in the step: the same bytes)
    from cflearn_amd.optim import FusedAdam, ParamArena

    model = C.vit_b16_classifier(1000).to(dev)
    arena = ParamArena([p for p in model.parameters()], with_shadow=True)
    opt = FusedAdam(None, lr=1e-4, arena=arena)
    arena.flat_g.normal_()
    nparam = arena.flat_p.numel()

    def adam() -> None:
        opt.prepare_step()
        opt.launch_step()

    measure(f"fused AdamW, {nparam / 1e6:.1f} M parameters (one launch)", "hbm", 1, adam, 0, nparam * 30.0)

    use = [r for r in rows if not (r["name"].startswith("LayerNorm backward") and (("two-word" in r["name"]) != (words == 2)))]
    total = sum(r["j"] * r["count"] for r in use)
    by_cls = {}
    for r in use:
        by_cls[r["cls"]] = by_cls.get(r["cls"], 0.0) + r["j"] * r["count"]
    out()
    out(f"# sum over the step's launches (LayerNorm backward: the {words}-word row): {total:.2f} J / step above idle = {100 * total / j_step:.1f} % of the step's "
        f"{j_step:.2f} J  (" + ", ".join(f"{k} {v:.2f} J" for k, v in sorted(by_cls.items(), key=lambda kv: -kv[1])) + ")")
    t_alone = sum(r["us"] * r["count"] for r in use) * 1e-3
    out(f"# the same launches alone take {t_alone:.2f} ms (the step: {sec * 1e3:.2f} ms on three queues)")
    gem = sorted([r for r in use if r["flops"]], key=lambda r: -(r["j"] / r["flops"]))
    out("# joules per TFLOP, worst first (GEMM-class and attention rows):")
    for r in gem[:6]:
        out(f"    {r['name']:64s} {r['j'] / r['flops'] * 1e12:6.3f} J/TFLOP   {r['j'] * r['count']:6.3f} J/step   ({r['flops'] / r['us'] / 1e6:6.0f} TFLOP/s alone at {r['sclk'] or 0:.0f} MHz)")
    best = min(r["j"] / r["flops"] for r in gem)
    out(f"#   (best row: {best * 1e12:.3f} J/TFLOP; a step of {bench.FLOP_PER_SAMPLE * B / 1e12:.2f} TFLOP at that rate would spend {best * bench.FLOP_PER_SAMPLE * B:.2f} J)")
    out("# what the step's joules are spent on, largest first:")
    for r in sorted(use, key=lambda r: -r["j"] * r["count"])[:8]:
        out(f"    {r['name']:64s} {r['j'] * r['count']:6.3f} J/step = {100 * r['j'] * r['count'] / j_step:5.1f} %")
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")
        with open(os.path.splitext(args.out)[0] + ".json", "w") as f:
            json.dump(dict(idle=idle, step=dict(ms=sec * 1e3, telemetry=st, joules_above_idle=j_step), rows=rows, grad_stream_words=words), f, indent=1)
    tel.stop()


if __name__ == "__main__":
    main()
