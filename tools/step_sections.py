#!/usr/bin/env python3
"""GPU time of the sections of the headline training step WITHOUT a profiler attached (rocprofv3 slows the host by
~20 % and makes the serial middle look host-bound): hipEvents on the main stream at
  t0 step start | t1 encoders + IMU net done (features joined) | t2 loss computed | t3 tail backward done (feature
  gradients available) | t4 optimizer step issued
and the host time at which each of those points was ISSUED."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfg = make_config(seq=2)
torch.manual_seed(1)
ts = TrainStep(cfg, (5, 64, 2048), dev, 8)
batch = bench.synth_batch(1234, 8, 2, 5, 64, 2048, 50, dev)
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    marks.append((name, e, time.perf_counter()))


ff, tail = ts.model.forward_features, ts._tail


def ff_w(*a, **k):
    mark("t0")
    r = ff(*a, **k)
    mark("t1")
    if torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing() and r["lidar"] is not None and r["lidar"][0].requires_grad:
        r["lidar"][0].register_hook(lambda g: mark("t3"))
    return r


def tail_w(*a, **k):
    r = tail(*a, **k)
    if not torch.cuda.is_current_stream_capturing():
        mark("t2")
    return r


ts.model.forward_features, ts._tail = ff_w, tail_w
for _ in range(6):
    ts.step(*batch)
torch.cuda.synchronize()
acc = {}
N = 20
for _ in range(N):
    marks.clear()
    ts.step(*batch)
    mark("t4")
    torch.cuda.synchronize()
    d = {n: (e, h) for n, e, h in marks}
    e0, h0 = d["t0"]
    for n in ("t1", "t2", "t3", "t4"):
        if n in d:
            g, h = acc.get(n, (0., 0.))
            acc[n] = (g + e0.elapsed_time(d[n][0]), h + (d[n][1] - h0) * 1e3)
for n in ("t1", "t2", "t3", "t4"):
    if n in acc:
        print("%s  gpu %.2f ms   host issued at %.2f ms" % (n, acc[n][0] / N, acc[n][1] / N))
