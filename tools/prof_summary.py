"""Compact per-step kernel table from a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv)."""
import csv, re, sys
path, steps = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = 0.0
out = []
for r in rows:
    name = r["Name"]
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)[:86]
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms
    out.append((ms, name, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3))
print(f"{'kernel':86s} {'ms/step':>8s} {'calls':>6s} {'avg us':>8s}")
for ms, name, calls, avg in out:  # (every row: round 5's files were cut at 26 and hid the kernels a round was about)
    print(f"{name:86s} {ms:8.3f} {calls:6.1f} {avg:8.1f}")
print(f"{'TOTAL (all kernels)':86s} {tot:8.3f} {sum(c for _, _, c, _ in out):6.0f} launches per step")
