#!/usr/bin/env python3
"""GPU time of each PointSeg block of encoder 1 inside the LIVE overlapped training step (steady state: steps issued back to
back, no sync in between, no profiler): hipEvents at the block boundaries of the forward pass (the yields of
PSEncoder.forward_steps) and, through pre-hooks on the nodes that produced the block outputs (tensor hooks would switch the
lazy pool gradients off: functional._LAZY), of the backward pass.
usage: python tools/block_times.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
from deeplio_amd.nets import PS_BLOCKS
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ts = TrainStep(make_config(seq=2), (5, 64, 2048), dev, 8)
batch = bench.synth_batch(1234, 8, 2, 5, 64, 2048, 50, dev)
enc = ts.model.lidar_feat_net.encoder1
fs = enc.forward_steps
rec, on = [], [False]


def ev(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    rec.append((name, e))


def fs_w(x):
    # one yield per module: stem(+pool1), then per block its Fires and the SELayer(+pool)
    names = ["stem+pool1"]
    for name, fires, se, pool in PS_BLOCKS:
        names += ["%s.fire%d" % (name[5:], i) for i in range(len(fires))] + (["%s.se%s" % (name[5:], "+pool" if pool else "")] if se else [])
    if on[0]:
        ev("fwd start")
    for i, y in enumerate(fs(x)):
        if on[0] and i < len(names):
            ev("fwd " + names[i])
            t = y[0] if isinstance(y, tuple) else y
            if torch.is_tensor(t) and t.requires_grad and t.grad_fn is not None:
                t.grad_fn.register_prehook(lambda g, n=names[i]: ev("bwd: gradient of the output of " + n))
        yield y


enc.forward_steps = fs_w
for _ in range(8):
    ts.step(*batch)
on[0] = True
runs = []
for _ in range(10):
    rec.clear()
    ts.step(*batch)
    runs.append(list(rec))
torch.cuda.synchronize()
on[0] = False
names = [n for n, _ in runs[0]]
acc = [0.0] * len(names)
for r in runs:
    e0 = r[0][1]
    for i, (n, e) in enumerate(r):
        acc[i] += e0.elapsed_time(e)
prev = 0.0
for n, a in zip(names, acc):
    t = a / len(runs)
    print("%-46s at %7.2f ms   (+%.2f)" % (n, t, t - prev))
    prev = t
