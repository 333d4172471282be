"""Is the eager step CPU-launch bound?  enqueue time (host returns) vs completion time per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cflearn_amd as C
from cflearn_amd.engine import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda")
torch.manual_seed(0)
model = C.vit_b16_classifier(1000).to(dev)
ts = TrainStep(model, lr=1e-4, use_graph=False)
g = torch.Generator().manual_seed(1234)
img = torch.randn(B, 3, 224, 224, generator=g).to(dev)
labels = torch.randint(0, 1000, (B,), generator=g).to(dev)
for _ in range(5):
    ts.step(img, labels)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    ts.step(img, labels)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"batch {B}: host enqueue {(t1 - t0) / N * 1e3:.2f} ms/step, completion {(t2 - t0) / N * 1e3:.2f} ms/step")
# one isolated step: host time with an idle GPU queue
torch.cuda.synchronize()
t0 = time.perf_counter(); ts.step(img, labels); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"single step: host {1e3 * (t1 - t0):.2f} ms, done {1e3 * (t2 - t0):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    ts.step(img, labels)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
