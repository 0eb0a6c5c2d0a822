"""soak: K training steps of a configuration, the cooperative BatchNorm error words read every 50 steps (no spin-limit hit allowed)
usage: python tools/soak_cfg.py <steps> [bench_cfg flags...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, argparse
from deeplio_amd import ops
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
ap = argparse.ArgumentParser()
ap.add_argument("steps", type=int)
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
ts.check_every = 0
batch = bench.synth_batch(1, a.batch, a.seq, a.channels, 64, 2048, 50, dev)
bad, t0 = 0, time.perf_counter()
for i in range(0, a.steps, 50):
    for _ in range(50):
        ts.step(*batch)
    torch.cuda.synchronize()
    e = ops.bn_coop_errors()
    if e:
        bad += 1
        print("steps %d-%d: %d workspace(s) hit the spin limit" % (i, i + 49, e), flush=True)
        ops.bn_coop_check(fallback=False)
print("%s %s: %d steps, %.2f ms/step, chunks with a spin-limit hit: %d, mode %d" % (
    a.lidar, ov, a.steps, 1e3 * (time.perf_counter() - t0) / a.steps, bad, ops.lib.dlio_bn_coop_get_mode()))
