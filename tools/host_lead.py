#!/usr/bin/env python3
"""Where in the step does the GPU wait for the host?  Every autograd Function of deeplio_amd.functional gets a mark at the
start of its forward and backward: a hipEvent on the stream current there + the host clock.  lead = (GPU time at which the
stream reaches the mark) - (host time at which the mark was issued); a lead near zero means the stream had run dry and the
kernel behind the mark starts when the host gets to it (host-bound); no profiler attached.
CAVEAT (measured): the marks themselves cost the host ~10 us each and stretch the step by ~3 ms, which is what makes the
deep layers' backward look dry; without them the lead never reaches zero -- issuing the IMU branch later in forward
(25.3 vs 24.2 ms) or in backward (24.25 vs 24.25 ms) buys nothing.  Use it for ORDER, not for magnitudes.
usage: python tools/host_lead.py [--st]   (--st: single-threaded autograd)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from deeplio_amd import functional as Fh
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if "--st" in sys.argv:
    torch.autograd.set_multithreading_enabled(False)
cfg = make_config(seq=2)
torch.manual_seed(1)
ts = TrainStep(cfg, (5, 64, 2048), dev, 8)
batch = bench.synth_batch(1234, 8, 2, 5, 64, 2048, 50, dev)
marks, on = [], [False]


def mark(name):
    if not on[0]:
        return
    e = torch.cuda.Event(enable_timing=True)
    s = torch.cuda.current_stream()
    e.record(s)
    marks.append((name, e, time.perf_counter(), s.cuda_stream))


def wrap(cls):
    f, b = cls.forward, cls.backward

    def fw(ctx, *a, **k):
        mark(cls.__name__ + ".fwd")
        return f(ctx, *a, **k)

    def bw(ctx, *a, **k):
        mark(cls.__name__ + ".bwd")
        return b(ctx, *a, **k)
    cls.forward, cls.backward = staticmethod(fw), staticmethod(bw)


for n in dir(Fh):
    c = getattr(Fh, n)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c)
for _ in range(6):
    ts.step(*batch)
torch.cuda.synchronize()
on[0] = True
e0 = torch.cuda.Event(enable_timing=True); e0.record(torch.cuda.current_stream()); h0 = time.perf_counter()
ts.step(*batch)
mark("end")
torch.cuda.synchronize()
streams = {}
print("%-28s %8s %8s %8s  stream" % ("mark", "host ms", "gpu ms", "lead"))
dry = 0.0
for i, (n, e, h, s) in enumerate(marks):
    sid = streams.setdefault(s, len(streams))
    g = e0.elapsed_time(e); hh = (h - h0) * 1e3
    print("%-28s %8.2f %8.2f %8.2f  %d%s" % (n, hh, g, g - hh, sid, "   <- dry" if g - hh < 0.05 else ""))
