"""whole model: gradient agreement with the fp32 oracle of (a) the HIP bf16 path, (b) the HIP fp32 path, (c) the
CPU oracle under torch.autocast(bfloat16) -- PyTorch's own mixed-precision execution of the same model"""
import os, sys, types
import numpy as np
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_common as gc
from deeplio_amd import nets, losses, misc
from deeplio_amd.config import make_config
from deeplio_amd.se3 import se3_to_SE3
from oracle import model as om
from oracle import se3 as ose3
dev = torch.device("cuda:0")
H, W, B, S = 16, 1024, 2, 2


def cfg_of(prec):
    ov = dict(gc.NO_DROP); ov.update(gc.SMALL_RNN); ov['lidar-feat-pointseg/precision'] = prec
    return make_config(seq=S, overrides=ov)


batch = gc.make_batch(2000, B, S, 5, H, W, 7)


def hip(prec):
    cfg = cfg_of(prec)
    misc.build_config_container(cfg, types.SimpleNamespace(device="cuda:0", batch_size=B))
    m = nets.get_model((5, H, W), cfg, dev); gc.fill_state(m, 1000); m.train()
    crit = losses.get_loss_function(cfg, dev)
    xyz, nrm, imu, f2f, f2g = (t.to(dev) for t in batch)
    pt, pw = m([[xyz, nrm], imu]); pp, pq = se3_to_SE3(pt, pw)
    loss = crit(pt, pw, pp[:, 1:3], pq[:, 1:3], f2f[:, :, 0:3], f2f[:, :, 3:], f2g[:, 1:3, 0:3], f2g[:, 1:3, 3:7])
    loss.backward()
    return float(loss), {k: p.grad.detach().cpu().double() for k, p in m.named_parameters() if p.grad is not None}


def oracle(autocast, dtype=torch.float32):
    cfg = cfg_of('fp32')
    m = om.get_model((5, H, W), cfg); gc.fill_state(m, 1000); m = m.to(dtype).train()
    crit = om.get_loss_function(cfg).to(dtype)
    xyz, nrm, imu, f2f, f2g = (t.to(dtype) for t in batch)
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        pt, pw = m([[xyz, nrm], imu])
    pt, pw = pt.to(dtype), pw.to(dtype)
    pp, pq = ose3.se3_to_SE3(pt, pw)
    loss = crit(pt, pw, pp[:, 1:3], pq[:, 1:3], f2f[:, :, 0:3], f2f[:, :, 3:], f2g[:, 1:3, 0:3], f2g[:, 1:3, 3:7])
    loss.backward()
    return float(loss), {k: p.grad.detach().double() for k, p in m.named_parameters() if p.grad is not None}


def report(name, g, ref):
    big = [k for k in ref if "encoder" in k and k.endswith("weight") and ref[k].dim() == 4]
    cos = [float(F.cosine_similarity(g[k].flatten(), ref[k].flatten(), dim=0)) for k in big]
    print("%-28s encoder conv weights: cos median %.3f min %.3f max %.3f" % (name, np.median(cos), min(cos), max(cos)))


l64, g64 = oracle(False, torch.float64)
for name, (l, g) in (("HIP fp32", hip("fp32")), ("HIP bf16", hip("bf16")), ("oracle fp32", oracle(False)),
                     ("oracle autocast bf16", oracle(True))):
    print(name, "loss", l, "fp64 loss", l64)
    report(name, g, g64)
