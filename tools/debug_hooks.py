"""debug helper: compare grad_output / output of every named sub-module between HIP model and oracle"""
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
_, om, oc, ob = build_oracle(name, True)
rec = {}
def hook(tag, store):
    def f(mod, gin, gout):
        if gout and gout[0] is not None:
            store[tag] = gout[0].detach().double().cpu()
    return f
def fhook(tag, store):
    def f(mod, inp, out):
        if torch.is_tensor(out):
            store[tag] = out.detach().double().cpu()
    return f
H, O, HF, OF = {}, {}, {}, {}
for (k, m), (k2, m2) in zip(model.named_modules(), om.named_modules()):
    pass
hm = dict(model.named_modules()); omm = dict(om.named_modules())
for k in hm:
    if k in omm and k.count(".") <= 3 and k:
        hm[k].register_full_backward_hook(hook(k, H)); omm[k].register_full_backward_hook(hook(k, O))
        hm[k].register_forward_hook(fhook(k, HF)); omm[k].register_forward_hook(fhook(k, OF))
pt, pw, pp, pq, loss = hip_step_forward(model, crit, batch)
loss.backward()
xyz, nrm, imu, gt_f2f, gt_f2g = ob
a, b = om([[xyz, nrm], imu])
p2, q2 = ose3.se3_to_SE3(a, b)
ol = oc(a, b, p2[:, 1:3], q2[:, 1:3], gt_f2f[:, :, 0:3], gt_f2f[:, :, 3:], gt_f2g[:, 1:3, 0:3], gt_f2g[:, 1:3, 3:7])
ol.backward()
def rel(x, y):
    if x.shape != y.shape: return float("nan")
    return float((x - y).abs().max()) / max(float(y.abs().max()), 1e-30)
for k in hm:
    if k in H and k in O:
        print("%-55s fwd %.1e  gout %.1e  %s" % (k, rel(HF[k], OF[k]) if k in HF and k in OF else -1, rel(H[k], O[k]), tuple(O[k].shape)))
