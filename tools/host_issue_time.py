"""How long does the host take to ISSUE one training step (no sync) vs. the GPU to execute it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep

dev = torch.device("cuda", 0)
cfg = make_config(seq=2)
ts = TrainStep(cfg, (5, 64, 2048), dev, 8)
batch = bench.synth_batch(1, 8, 2, 5, 64, 2048, 50, dev)
for _ in range(3):
    ts.step(*batch)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    ts.step(*batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("issue %.1f ms, total %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t0)))
