#!/usr/bin/env python3
"""Isolated timing of the four BatchNorm launches (statistics, apply, backward reduce, backward
apply) on every PointSeg conv output at the headline shape (N=16 images per encoder); prints us and
TB/s on the algorithmic bytes (1, 2, 2 and 3 tensor passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
N = 16


COLD = "--cold" in sys.argv      # evict L2 / Infinity Cache between calls (the in-step condition)
CLEAN = "--clean" in sys.argv    # evict with reads (no dirty lines left to write back during the timed call)
_flush = torch.zeros(1 << 28, device=dev) if COLD else None      # 1 GiB


def timeit(fn, iters=10):
    if COLD:
        t = 0.0
        for _ in range(4):
            _flush.sum() if CLEAN else _flush.add_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            t += a.elapsed_time(b)
        return t / 4 * 1e3
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


shapes = [("stem", 64, 64, 64, 1024)]
H, W = 64, 512
blocks = [("b1", [(16, 64)] * 2, (1, 2)), ("b2", [(32, 128)] * 2, (1, 2)), ("b3", [(48, 192)] * 2 + [(64, 256)] * 2, (2, 2)),
          ("b4", [(64, 256)] * 2, (2, 2)), ("b5", [(80, 384)] * 2, None)]
for bn, fires, pool in blocks:
    for i, (sq, e) in enumerate(fires):
        shapes += [("%s.%d.sq" % (bn, i), sq, sq, H, W), ("%s.%d.e" % (bn, i), e, 2 * e, H, W)]
    if pool:
        H, W = H // pool[0], W // pool[1]
tot = [0.0] * 4
print("%-9s %4s %9s | %7s %5s | %7s %5s | %7s %5s | %7s %5s" % ("tensor", "C", "HxW", "stats", "TB/s", "apply", "TB/s", "bwdred", "TB/s", "bwdapp", "TB/s"))
for name, C, ctot, H, W in shapes:
    HW = H * W
    mult = 2 if name.endswith(".e") else 1        # e1 and e3 halves of the concat buffer: two BN launches
    x = torch.randn(N, ctot, H, W, device=dev); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev); rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    ws = ops._stats_ws(N, C, HW, dev)
    prm = ops.bn_train_apply(x, ctot, 0, g, b, 1e-5, 0.1, rm, rv, y, ctot, 0, N, C, HW, False, True)
    lib, _ptr, _stream = ops.lib, ops._ptr, ops._stream

    def fwd(phase):
        ops.check(lib.dlio_bn_train_apply(_ptr(x), N, ctot, 0, C, HW, 0, 1, _ptr(g), _ptr(b), 1e-5, 0.1, _ptr(rm), _ptr(rv),
                                          _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]), None, 0, 0, _ptr(y), ctot, 0, None, 0, 0,
                                          _ptr(ws), ws.numel(), phase, 1.0, None, None, None, None, _stream()), "fwd")

    def bwd(phase):
        ops.check(lib.dlio_bn_bwd(_ptr(dy), ctot, 0, _ptr(x), ctot, 0, _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]), _ptr(b),
                                  _ptr(dx), ctot, 0, _ptr(dg), _ptr(db), 0, N, C, HW, 0, 1, 1, _ptr(ws), ws.numel(), phase, 1.0,
                                  None, None, _stream()), "bwd")
    ts = [timeit(lambda: fwd(1)), timeit(lambda: fwd(2)), timeit(lambda: bwd(1)), timeit(lambda: bwd(2))]
    byt = 4.0 * N * C * HW
    passes = [1, 2, 2, 3]
    print("%-9s %4d %4dx%-4d | " % (name, C, H, W) + " | ".join("%7.1f %5.2f" % (t, byt * k / t / 1e6) for t, k in zip(ts, passes)))
    for i in range(4): tot[i] += ts[i] * mult
print("per encoder (us):", [round(t, 1) for t in tot], " x2 encoders = %.2f ms/step" % (2 * sum(tot) / 1e3))
