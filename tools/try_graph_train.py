"""experiment: replay the whole training step as one hipGraph (semantics of dropout / Adam step counter
are NOT graph-safe yet -- timing only)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
dev = torch.device("cuda", 0)
ts = TrainStep(make_config(seq=2), (5, 64, 2048), dev, 8)
batch = bench.synth_batch(1, 8, 2, 5, 64, 2048, 50, dev)
for _ in range(3): ts.step(*batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ts.step(*batch)
torch.cuda.synchronize(); print("eager: %.2f ms/step" % ((time.perf_counter() - t0) / 5 * 1e3))
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): ts.step(*batch)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = ts.step(*batch)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize(); print("graph replay: %.2f ms/step, loss %.4f" % ((time.perf_counter() - t0) / 10 * 1e3, float(loss)))
