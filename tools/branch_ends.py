#!/usr/bin/env python3
"""When do the branches of the forward pass END on the GPU?  hipEvents behind the IMU net (its own stream) and behind the
two encoders (joined on the main stream), relative to the step start; no profiler.  usage: python tools/branch_ends.py [--dtype bf16]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
bf16 = "bf16" in sys.argv
S = 4 if bf16 else 2
cfg = make_config(seq=S)
if bf16:
    cfg['lidar-feat-pointseg']['precision'] = 'bf16'
ts = TrainStep(cfg, (5, 64, 2048), dev, 8)
batch = bench.synth_batch(1234, 8, S, 5, 64, 2048, 50, dev)
ev = {}


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    ev[name] = (e, time.perf_counter())


m = ts.model
imu_f, enc_f, ff = m.imu_feat_net.forward, m.lidar_feat_net.encode, m.forward_features


def imu_w(*a, **k):
    mark("imu issue start")
    r = imu_f(*a, **k)
    mark("imu end")
    return r


def enc_w(*a, **k):
    mark("enc issue start")
    r = enc_f(*a, **k)
    mark("enc end")
    return r


def ff_w(*a, **k):
    mark("t0")
    return ff(*a, **k)


m.imu_feat_net.forward, m.lidar_feat_net.encode, m.forward_features = imu_w, enc_w, ff_w
for _ in range(6):
    ts.step(*batch)
torch.cuda.synchronize()
acc = {}
N = 10
for _ in range(N):
    ev.clear()
    ts.step(*batch)
    mark("step end")
    torch.cuda.synchronize()
    e0, h0 = ev["t0"]
    for n, (e, h) in ev.items():
        g, hh = acc.get(n, (0., 0.))
        acc[n] = (g + e0.elapsed_time(e), hh + (h - h0) * 1e3)
for n, (g, h) in acc.items():
    print("%-18s gpu %6.2f ms   host %6.2f ms" % (n, g / N, h / N))
