#!/usr/bin/env python3
"""Is the host on the step's critical path at the serial middle?  A busy-wait of X us is inserted on the host right behind
the tail forward (before loss.backward()); if the step grows by ~X the GPU was waiting for the host there, if it does not
grow the host had that much lead.  usage: python tools/host_delay_probe.py [us ...]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ts = TrainStep(make_config(seq=2), (5, 64, 2048), dev, 8)
batch = bench.synth_batch(1234, 8, 2, 5, 64, 2048, 50, dev)
delay = [0.0]
tail = ts._tail


def tail_w(*a, **k):
    r = tail(*a, **k)
    t = time.perf_counter() + delay[0]
    while time.perf_counter() < t:
        pass
    return r


ts._tail = tail_w
for _ in range(8):
    ts.step(*batch)
for us in [int(a) for a in sys.argv[1:]] or [0, 250, 500, 1000, 2000, 0]:
    delay[0] = us * 1e-6
    for _ in range(3):
        ts.step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ts.step(*batch)
    torch.cuda.synchronize()
    print("host delay %5d us -> %.3f ms/step" % (us, (time.perf_counter() - t0) * 50))
