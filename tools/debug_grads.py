"""debug helper: per-parameter gradient error of the HIP path vs the oracle for one golden case"""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_common as gc
from test_gpu_model import build, build_oracle, hip_step_forward
from oracle import se3 as ose3
name = sys.argv[1]
dev = torch.device("cuda:0")
_, model, crit, batch = build(name, dev, True)
pt, pw, pp, pq, loss = hip_step_forward(model, crit, batch)
loss.backward()
_, om, oc, ob = build_oracle(name, True)
xyz, nrm, imu, gt_f2f, gt_f2g = ob
a, b = om([[xyz, nrm], imu])
p2, q2 = ose3.se3_to_SE3(a, b)
ol = oc(a, b, p2[:, 1:3], q2[:, 1:3], gt_f2f[:, :, 0:3], gt_f2f[:, :, 3:], gt_f2g[:, 1:3, 0:3], gt_f2g[:, 1:3, 3:7])
ol.backward()
op = dict(om.named_parameters())
print("loss", float(loss), float(ol))
for k, p in model.named_parameters():
    og = op[k].grad
    if og is None or p.grad is None:
        print("%-70s none" % k); continue
    x, y = p.grad.double().cpu(), og.double()
    print("%-70s rel %.2e  max %.2e" % (k, float((x - y).abs().max()) / max(float(y.abs().max()), 1e-30), float(y.abs().max())))
