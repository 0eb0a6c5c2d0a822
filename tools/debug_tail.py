#!/usr/bin/env python3
"""eager step vs tail-graph step: first diverging quantity (loss, flat gradient per parameter)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_common as gc
from deeplio_amd import functional as Fh
from deeplio_amd.config import make_config
from deeplio_amd.trainer import TrainStep
dev = torch.device("cuda:0")
cfg = make_config(seq=2)
batches = [tuple(t.to(dev) for t in gc.make_batch(20 + i, 2, 2, 5, 64, 256, 50)) for i in range(3)]
out = []
for graph in (False, True):
    torch.manual_seed(3)
    ts = TrainStep(cfg, (5, 64, 256), dev, 2)
    ts.tail_mode, ts.tail_after = graph, 2
    Fh.manual_seed(9)
    rec = []
    orig_step = ts.optimizer.step
    def spy(*a, **k):
        rec.append(ts.optimizer.grad.clone())
        return orig_step(*a, **k)
    ts.optimizer.step = spy
    losses = [ts.step(*batches[i % 3]).clone() for i in range(4)]
    torch.cuda.synchronize()
    out.append((losses, rec, ts))
(la, ga, tsa), (lb, gb, tsb) = out
names = [n for n, _ in list(tsa.model.named_parameters()) + [("crit." + n, p) for n, p in tsa.criterion.named_parameters()]]
opt = tsa.optimizer
for i in range(4):
    print("step", i + 1, "loss", float(la[i]), float(lb[i]), "equal", torch.equal(la[i], lb[i]), "grads equal", torch.equal(ga[i], gb[i]))
    if not torch.equal(ga[i], gb[i]):
        bad = 0
        for p, o, n in zip(opt.params, opt.offsets, names):
            a, b = ga[i][o:o + p.numel()], gb[i][o:o + p.numel()]
            if not torch.equal(a, b):
                bad += 1
                if bad <= 12:
                    print("   ", n, tuple(p.shape), "rel", float((a - b).norm() / (a.norm() + 1e-30)), "|a|", float(a.norm()), "|b|", float(b.norm()))
        print("   ", bad, "of", len(names), "parameters differ")
        break
