"""Which python lines issue aten::copy_ / clone / contiguous / fill during one training step?"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
dev = torch.device("cuda", 0)
cfg = make_config(lidar="lidar-feat-pointseg", imu="imu-feat-rnn", fusion="fusion-layer-soft", odom="odom-feat-rnn", seq=2)
ts = TrainStep(cfg, (5, 64, 256), dev, 2)
batch = bench.synth_batch(1, 2, 2, 5, 64, 256, 50, dev)
for _ in range(2): ts.step(*batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    ts.step(*batch)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::add_", "aten::add", "aten::cat", "aten::stack", "aten::mul", "aten::sum",
                  "aten::index", "aten::select_backward", "aten::slice_backward", "aten::zeros_like", "aten::neg"):
        st = [f for f in (e.stack or []) if "deeplio_amd" in f or "bench.py" in f]
        cnt[(e.name, " <- ".join(x.split("/")[-1] for x in st[:2]) if st else "(engine)", str(e.input_shapes)[:60])] += 1
for k, v in cnt.most_common(60): print(v, k)
