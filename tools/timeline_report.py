"""Per-queue view of ONE training step from a reduced kernel trace (tools/trace_reduce.py output):
busy time, gaps and the forward / backward split on every hardware queue.

    python tools/timeline_report.py gpurun_out/<tag>/timeline.csv.gz [step-index-from-end]"""
import csv
import gzip
import sys
from collections import defaultdict


def main() -> None:
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = list(csv.DictReader(gzip.open(path, "rt") if path.endswith(".gz") else open(path)))
    rows = [dict(q=r["queue"], name=r["name"], s=float(r["start_us"]), d=float(r["dur_us"])) for r in rows]
    # a step = from the first kernel of one forward (the patch embedding's im2row) to the next (round 4: the optimizer runs as
    # several range launches inside backward, so "the Adam kernel" no longer marks the end of a step); traces without an
    # im2row kernel fall back to the round-1..3 rule (end of one adam_dev_kernel to the end of the next)
    marks = [r for r in rows if r["name"].split("<")[0].endswith("im2row_kernel") and "conv_im2row" not in r["name"]]
    if len(marks) < back + 1:  # the UNet step: one diffusion-loss kernel per step
        marks = [r for r in rows if "diffusion_loss" in r["name"]]
    if len(marks) >= back + 1:
        t0, t1 = marks[-back - 1]["s"], marks[-back]["s"]
    else:
        adam = [r for r in rows if "adam_dev" in r["name"]]
        if len(adam) < back + 1:
            raise SystemExit("not enough steps in the trace")
        t0, t1 = adam[-back - 1]["s"] + adam[-back - 1]["d"], adam[-back]["s"] + adam[-back]["d"]
    step = [r for r in rows if t0 <= r["s"] < t1]
    print(f"step window {t1 - t0:.1f} us, {len(step)} kernels")
    xent = [r for r in step if "xent" in r["name"]]
    tb = xent[0]["s"] if xent else None
    if tb:
        print(f"forward (to the loss kernel) {tb - t0:.1f} us, backward + optimizer {t1 - tb:.1f} us")
    byq = defaultdict(list)
    for r in step:
        byq[r["q"]].append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: -sum(r["d"] for r in kv[1])):
        rs.sort(key=lambda r: r["s"])
        busy = sum(r["d"] for r in rs)
        gaps = [rs[i + 1]["s"] - (rs[i]["s"] + rs[i]["d"]) for i in range(len(rs) - 1)]
        big = sorted(((g, rs[i]["name"][:40], rs[i + 1]["name"][:40]) for i, g in enumerate(gaps) if g > 20), reverse=True)[:6]
        span = rs[-1]["s"] + rs[-1]["d"] - rs[0]["s"]
        print(f"queue {q}: {len(rs)} kernels, busy {busy:.0f} us over a span of {span:.0f} us (first at +{rs[0]['s'] - t0:.0f}), "
              f"sum of gaps {sum(g for g in gaps if g > 0):.0f} us")
        fam = defaultdict(float)
        for r in rs:
            fam[r["name"].replace("(anonymous namespace)::", "").split("<")[0]] += r["d"]
        print("    " + ", ".join(f"{k} {v:.0f}" for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:8]))
        for g, a, b in big:
            print(f"    gap {g:7.1f} us between {a} -> {b}")


if __name__ == "__main__":
    main()
