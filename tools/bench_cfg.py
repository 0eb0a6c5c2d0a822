"""informational: time a training step of an arbitrary configuration (overrides as key=value)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--lidar", default="lidar-feat-pointseg"); ap.add_argument("--imu", default="imu-feat-rnn")
ap.add_argument("--fusion", default="fusion-layer-soft"); ap.add_argument("--odom", default="odom-feat-rnn")
ap.add_argument("--channels", type=int, default=5); ap.add_argument("--seq", type=int, default=2)
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--set", nargs="*", default=[])
a = ap.parse_args()
ov = {}
for kv in a.set:
    k, v = kv.split("=", 1)
    try: v = json.loads(v)
    except ValueError: pass
    ov[k] = v
cfg = make_config(a.lidar, a.imu, a.fusion, a.odom, a.seq, overrides=ov)
dev = torch.device("cuda", 0)
ts = TrainStep(cfg, (a.channels, 64, 2048), dev, a.batch)
batch = bench.synth_batch(1, a.batch, a.seq, a.channels, 64, 2048, 50, dev)
for _ in range(3): ts.step(*batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ts.step(*batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("%s %s %s %s %s: %.1f ms/step, %.1f frame-pairs/s" % (a.lidar, a.imu, a.fusion, a.odom, ov, dt * 1e3, a.batch * a.seq / dt))
