#!/usr/bin/env python3
"""How long is the serial middle of the training step?  hipEvents on the main stream: after the
lidar feature net's forward (E1), when its output gradient arrives in backward (E2), step start /
end.  E1 -> E2 = fusion, odometry RNN, heads, SE(3) chain, loss and their backward: nothing of the
encoders can overlap it."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
dev = torch.device("cuda:0")
cfg = make_config(seq=2)
torch.manual_seed(1)
ts = TrainStep(cfg, (5, 64, 2048), dev, 8)
batch = bench.synth_batch(1234, 8, 2, 5, 64, 2048, 50, dev)
ev = {k: [torch.cuda.Event(enable_timing=True) for _ in range(12)] for k in ("s", "e1", "e2", "end")}
it = [0]
net = ts.model.lidar_feat_net
orig = net.forward


def fwd(x):
    y = orig(x)
    ev["e1"][it[0]].record()
    if y.requires_grad:
        y.register_hook(lambda g: ev["e2"][it[0]].record())
    return y


net.forward = fwd
for i in range(12):
    it[0] = i
    ev["s"][i].record()
    ts.step(*batch)
    ev["end"][i].record()
torch.cuda.synchronize()
f = lambda a, b: sum(ev[a][i].elapsed_time(ev[b][i]) for i in range(4, 12)) / 8
print("forward to lidar features %.2f ms | serial middle (fusion .. loss .. fusion backward) %.2f ms | "
      "encoder backward + Adam %.2f ms | step %.2f ms" % (f("s", "e1"), f("e1", "e2"), f("e2", "end"), f("s", "end")))
