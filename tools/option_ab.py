"""In-process A/B of one integer option of libcfhip.so (cfhip_set_option): per-shape GEMM times (bench.time_gemms) and the
whole training step, interleaved rounds.   python tools/option_ab.py <option> <v0,v1,...> [batch]"""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import cflearn_amd as C
from cflearn_amd import ops
from cflearn_amd.engine import TrainStep

OPT = sys.argv[1]
VALUES = [int(v) for v in sys.argv[2].split(",")]
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 128
bench.time_gemms(BATCH, 5)  # warm
for v in VALUES:
    ops.set_option(OPT, v)
    flops, tsec, rows, _ = bench.time_gemms(BATCH, 20)
    print(f"{OPT} {v}: {tsec * 1e3:.3f} ms GEMM / step  " + "  ".join(
        f"{r['layout']}{r['N']}x{r['K']}/{r['epilogue'][:5]}:{r['us']:.0f}" for r in rows), flush=True)
ops.set_option(OPT, VALUES[0])

dev = torch.device("cuda")
torch.manual_seed(0)
model = C.vit_b16_classifier(1000).to(dev)
ts = TrainStep(model, lr=1e-4, use_graph=False)
g = torch.Generator().manual_seed(1234)
img = torch.randn(BATCH, 3, 224, 224, generator=g).to(dev)
labels = torch.randint(0, 1000, (BATCH,), generator=g).to(dev)

def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step(img, labels)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

run(5)
res = {v: [] for v in VALUES}
for rnd in range(5):
    for v in VALUES:
        ops.set_option(OPT, v)
        res[v].append(run(10))
for v, r in res.items():
    print(f"step, {OPT} {v}: median {statistics.median(r):7.3f} ms  min {min(r):7.3f}  all {[round(x, 2) for x in r]}")
