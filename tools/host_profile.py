"""cProfile of the host side of the training step (launch issue, autograd bookkeeping): which Python
functions the 17 ms of host time per step go to.  usage: python tools/host_profile.py [--dtype bf16]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep

dev = torch.device("cuda", 0)
bf16 = "bf16" in sys.argv
S = 4 if bf16 else 2
cfg = make_config(seq=S)
if bf16:
    cfg['lidar-feat-pointseg']['precision'] = 'bf16'
ts = TrainStep(cfg, (5, 64, 2048), dev, 8)
batch = bench.synth_batch(1, 8, S, 5, 64, 2048, 50, dev)
for _ in range(5):
    ts.step(*batch)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    ts.step(*batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("issue %.2f ms/step, wall %.2f ms/step" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
if "--st" in sys.argv:                 # backward on the calling thread: cProfile sees the backward functions too
    torch.autograd.set_multithreading_enabled(False)
    for _ in range(3):
        ts.step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        ts.step(*batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("single-threaded autograd: issue %.2f ms/step, wall %.2f ms/step" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    ts.step(*batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(60)
