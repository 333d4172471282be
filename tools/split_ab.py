"""Whole-step A/B of the split-K factor of the dW GEMMs (ops.pick_split_k: ceil(target / tiles128) slices, target = 512 shipped):
python tools/split_ab.py [batch]"""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cflearn_amd as C
from cflearn_amd import ops
from cflearn_amd.engine import TrainStep

BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 128
TARGETS = [int(v) for v in os.environ.get("SPLIT_TARGETS", "512,256,384,768,1024").split(",")]
if os.environ.get("GEMM_HEURISTIC"):
    ops.set_option("gemm_heuristic", int(os.environ["GEMM_HEURISTIC"]))
STATE = {"target": 512}


def pick(m, n, k):
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    steps = (k + 63) // 64
    if tiles >= 256 or steps < 8:
        return 1
    return max(1, min(steps // 4, (STATE["target"] + tiles - 1) // tiles))


ops.pick_split_k = pick
import cflearn_amd.functional as HF, cflearn_amd.fused as FU
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
res = {t: [] for t in TARGETS}
for rnd in range(5):
    for t in TARGETS:
        STATE["target"] = t
        res[t].append(run(10))
for t, r in res.items():
    STATE["target"] = t
    print(f"target {t:5d} (768x3072: split {pick(768, 3072, 25216)}, 768x768: {pick(768, 768, 25216)}): median {statistics.median(r):7.3f} ms  min {min(r):7.3f}")
