#!/usr/bin/env python3
"""Where does the fp32 gradient noise of the headline model come from?  Runs the CPU oracle (torch) in fp32
and fp64 on the same weights / batch and prints, per module output of encoder 1 (in backward order), the
relative L2 distance between the fp32 and the fp64 gradient that arrives there.  CPU only (test tooling)."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import golden_common as gc
from deeplio_amd.config import make_config
from oracle import model as om, se3 as ose3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
cfg = make_config(seq=2, overrides=gc.NO_DROP)
batch = gc.make_batch(7, B, 2, 5, 64, W, 50)


def run(dtype):
    m = om.get_model((5, 64, W), cfg); gc.fill_state(m, 1000); m = m.to(dtype).train()
    c = om.get_loss_function(cfg).to(dtype)
    grads = {}
    def hook(name):
        def f(mod, gin, gout):
            grads[name] = gout[0].detach().double()
        return f
    for name, mod in m.named_modules():
        if name.startswith("lidar_feat_net.encoder1") and name.count(".") <= 3 and name != "lidar_feat_net.encoder1":
            mod.register_full_backward_hook(hook(name))
    for name in ("lidar_feat_net", "imu_feat_net", "fusion_net", "odom_feat_net"):
        getattr(m, name).register_full_backward_hook(hook(name))
    xyz, nrm, imu, f2f, f2g = (t.to(dtype) for t in batch)
    a, b = m([[xyz, nrm], imu]); p2, q2 = ose3.se3_to_SE3(a, b)
    lo = c(a, b, p2[:, 1:3], q2[:, 1:3], f2f[:, :, 0:3], f2f[:, :, 3:], f2g[:, 1:3, 0:3], f2g[:, 1:3, 3:7])
    lo.backward()
    return grads

g32, g64 = run(torch.float32), run(torch.float64)
for k in g64:
    e = float((g32[k] - g64[k]).norm() / g64[k].norm())
    print("%-55s %-28s rel-L2 %.2e" % (k, tuple(g64[k].shape), e))
