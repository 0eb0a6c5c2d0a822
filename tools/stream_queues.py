"""order in which the auxiliary streams of one training step are created (it decides which streams share a hardware queue)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from deeplio_amd import functional as Fh, ops
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
dev = torch.device("cuda:0")
cfg = make_config(lidar="lidar-feat-pointseg", imu="imu-feat-rnn", fusion="fusion-layer-soft", odom="odom-feat-rnn", seq=2)
_dummies = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("WATCH_DUMMY_STREAMS", "0")))]
for _s in _dummies:
    with torch.cuda.stream(_s):
        torch.zeros(1, device=dev)
ts = TrainStep(cfg, (5, 64, 2048), dev, 8)
batch = bench.synth_batch(1234, 8, 2, 5, 64, 2048, 50, dev)
print("default stream %x" % torch.cuda.current_stream().cuda_stream)
ts.step(*batch); torch.cuda.synchronize()
for k, s in Fh._AUX.items():
    print(k, "%x" % s.cuda_stream)


names = ["main"] + [k[1] for k in Fh._AUX]
streams = [torch.cuda.current_stream(dev)] + list(Fh._AUX.values())
cls = []
for i, s in enumerate(streams):
    for c in cls:
        if ops.streams_share_queue(streams[c[0]], s):
            c.append(i); break
    else:
        cls.append([i])
print("queue classes:", [[names[i] for i in c] for c in cls])
