"""per-step time + cooperative BatchNorm error flag, step by step (usage: env ... python tools/step_watch.py [steps])"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from deeplio_amd import ops
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
dev = torch.device("cuda:0")
C, H, W, T, S, B = 5, 64, 2048, 50, 2, 8
cfg = make_config(lidar="lidar-feat-pointseg", imu="imu-feat-rnn", fusion="fusion-layer-soft", odom="odom-feat-rnn", seq=S)
_dummies = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("WATCH_DUMMY_STREAMS", "0")))]   # rotates the stream -> hardware queue mapping
for _s in _dummies:
    with torch.cuda.stream(_s):
        torch.zeros(1, device=dev)
torch.manual_seed(20260928)
ts = TrainStep(cfg, (C, H, W), dev, B)
batch = bench.synth_batch(1234, B, S, C, H, W, T, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if len(sys.argv) > 3:
    ops.prof_sample(7); ops.prof_enable(sum(1 << k for k in (6, 7, 8, 9)))
tot, cnt, nerr = 0.0, 0, 0
for i in range(0, n, every):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(every):
        loss = ts.step(*batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3 / every
    err = ops.bn_coop_errors()
    nerr += 1 if err else 0
    if i >= every:
        tot += dt; cnt += 1
    if os.environ.get("WATCH_VERBOSE") and (err or dt > 60 or i < 3):
        print("step", i, "%.2f ms" % dt, "loss %.6f" % float(loss), "errors", err, [e[1][:4].tolist() for e in ops._COOP_WS.values()], flush=True)
    if err:
        ops.bn_coop_check(fallback=False)
print("avg %.3f ms/step  chunks with errors %d  peak memory %.2f GB" % (tot / max(cnt, 1), nerr, torch.cuda.max_memory_allocated() / 1e9))
