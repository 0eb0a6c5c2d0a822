"""How long does the host take to ISSUE one training step (no sync), phase by phase, against what the GPU needs to execute it?
At the headline size the host runs ahead wherever the kernels are long; at a tiny size (64x256, B = 2) the step IS the host time.
usage: python tools/host_issue_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deeplio_amd import functional as Fh
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep

dev = torch.device("cuda", 0)
for (W, B) in ((2048, 8), (256, 2)):
    cfg = make_config(lidar="lidar-feat-pointseg", imu="imu-feat-rnn", fusion="fusion-layer-soft", odom="odom-feat-rnn", seq=2)
    ts = TrainStep(cfg, (5, 64, W), dev, B)
    batch = bench.synth_batch(1, B, 2, 5, 64, W, 50, dev)
    imgs, normals, imus, gf, gg = batch
    for _ in range(5):
        ts.step(*batch)
    torch.cuda.synchronize()
    acc = [0.0] * 6
    R = 10
    for _ in range(R):
        torch.cuda.synchronize()
        t = [time.perf_counter()]
        ts._steps += 1
        Fh.lazy_clear()
        feats = ts.model.forward_features([[imgs, normals], imus]); t.append(time.perf_counter())
        loss = ts._tail(feats, gf, gg); t.append(time.perf_counter())
        ts.optimizer.zero_grad(); t.append(time.perf_counter())
        if ts.autograd_inline:
            with torch.autograd.set_multithreading_enabled(False):
                loss.backward()
        else:
            loss.backward()
        t.append(time.perf_counter())
        ts.optimizer.step(); t.append(time.perf_counter())
        torch.cuda.synchronize(); t.append(time.perf_counter())
        for i in range(6):
            acc[i] += (t[i + 1] - t[i]) * 1e3 / R
    print("64x%d B=%d: issue forward_features %.2f  tail %.2f  zero_grad %.2f  backward %.2f  optimizer %.2f  = %.2f ms;  "
          "then %.2f ms until the GPU is done (total %.2f)" % (W, B, acc[0], acc[1], acc[2], acc[3], acc[4], sum(acc[:5]), acc[5], sum(acc)))
    ts.release_gc()
