"""Why is the host-fed step slower than the resident one?  enqueue vs completion time, with and without the feeder."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cflearn_amd as C
from cflearn_amd.engine import TrainStep
from cflearn_amd.data import TensorBatcher

B = 128
dev = torch.device("cuda")
torch.manual_seed(0)
model = C.vit_b16_classifier(1000).to(dev)
ts = TrainStep(model, lr=1e-4)
g = torch.Generator().manual_seed(1)
host = [dict(input=torch.randn(B, 3, 224, 224, generator=g).numpy(), labels=torch.randint(0, 1000, (B,), generator=g).numpy()) for _ in range(4)]
img = torch.from_numpy(host[0]["input"]).to(dev); lab = torch.from_numpy(host[0]["labels"]).to(dev)

class Endless:
    def __len__(self): return 1 << 30
    def __iter__(self): return itertools.cycle(host)

def run(next_batch, n=15, tag=""):
    for _ in range(3): ts.step(*next_batch())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ts.step(*next_batch())
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{tag:34s} enqueue {(t1 - t0) / n * 1e3:6.2f} ms/step   completion {(t2 - t0) / n * 1e3:6.2f} ms/step", flush=True)

run(lambda: (img, lab), tag="resident")
feed = iter(TensorBatcher(Endless(), dev, depth=1))
def nb():
    b = next(feed); return b["input"], b["labels"]
run(nb, tag="TensorBatcher (copy stream, 1 ahead)")
# synchronous pageable copy inside the step (what the reference batcher does)
def ref_style():
    h = host[0]; return torch.from_numpy(h["input"]).to(dev), torch.from_numpy(h["labels"]).to(dev)
run(ref_style, tag="reference style (.to(device) in-step)")
