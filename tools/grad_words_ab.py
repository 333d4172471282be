import sys, time, torch
sys.path.insert(0, '/root/repo')
import cflearn_amd as C
from cflearn_amd import fused
from cflearn_amd.engine import TrainStep
dev = torch.device("cuda")
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
ring = [(torch.randn(128,3,224,224,generator=g).to(dev), torch.randint(0,1000,(128,),generator=g).to(dev)) for _ in range(4)]
res = {}
for rep in range(3):
    for words in (1, 2):
        fused.GRAD_STREAM_WORDS = words
        fused._plans.clear()
        m = C.vit_b16_classifier(1000).to(dev)
        ts = TrainStep(m, lr=1e-4)
        for i in range(6): ts.step(*ring[i % 4])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(30): ts.step(*ring[i % 4])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30 * 1e3
        res.setdefault(words, []).append(round(dt, 3))
        del ts, m; fused._plans.clear(); torch.cuda.empty_cache()
        print(f"rep {rep} words {words}: {dt:.3f} ms/step", flush=True)
print(res)
