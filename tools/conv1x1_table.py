#!/usr/bin/env python3
"""Isolated timing of every 1x1 convolution launch of a PointSeg step (forward + data gradient) through the
routing functional.py uses (fp32-MFMA float4 / direct / split-K kernels, split-bf16 1x1 kernel), headline shape
(N = 16 images per encoder).  us and TB/s on input + output bytes; --cold evicts the caches between calls."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
from deeplio_amd import functional as Fh
dev = torch.device("cuda:0")
N = 16
COLD = "--cold" in sys.argv
_flush = torch.zeros(1 << 27, device=dev) if COLD else None


def timeit(fn, iters=10):
    if COLD:
        t = 0.0
        for _ in range(5):
            _flush.sum()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            t += a.elapsed_time(b)
        return t / 5 * 1e3
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


H, W = 64, 512
blocks = [("b1", [(64, 16, 64), (128, 16, 64)], (1, 2)), ("b2", [(128, 32, 128), (256, 32, 128)], (1, 2)),
          ("b3", [(256, 48, 192), (384, 48, 192), (384, 64, 256), (512, 64, 256)], (2, 2)),
          ("b4", [(512, 64, 256), (512, 64, 256)], (2, 2)), ("b5", [(512, 80, 384), (768, 80, 384)], None)]
tot_t = tot_b = 0.0
print("%-10s %-6s %5s %5s %9s %8s %7s  %s" % ("layer", "dir", "cin", "cout", "pixels", "us", "TB/s", "kernel"))
for bn, fires, pool in blocks:
    for i, (ci, sq, e) in enumerate(fires):
        for nm, a, b in (("sq", ci, sq), ("e1", sq, e)):
            for direction, cin, cout in (("fwd", a, b), ("dgrad", b, a)):
                x = torch.randn(N, cin, H, W, device=dev)
                y = torch.empty(N, cout, H, W, device=dev)
                w = torch.randn((b, a, 1, 1), device=dev) * 0.05
                mode = 0 if direction == "fwd" else 1
                d = ops.conv_desc(N, cin, H, W, cout, 1, 1, 1, 1, 0, 0)
                bx3 = Fh._use_bx3(N, cin, cout, 1, 1, (1, 1), H, W)
                if bx3:
                    wt = ops.conv1x1_bx3_prep(w, mode)
                    fn = lambda: ops.conv1x1_bx3_fwd(x, wt, None, y, d)
                else:
                    wt = ops.conv2d_prep_weight(w, mode)
                    fn = lambda: ops.conv2d_fwd(x, wt, None, y, d)
                t = timeit(fn)
                byt = 4.0 * N * H * W * (cin + cout)
                tot_t += t; tot_b += byt
                print("%-10s %-6s %5d %5d %9d %8.1f %7.2f  %s" % ("%s.%d.%s" % (bn, i, nm), direction, cin, cout, N * H * W, t,
                                                               byt / t / 1e6, "bx3" if bx3 else "fp32"))
    if pool:
        H, W = H // pool[0], W // pool[1]
print("total per encoder %.1f us, %.2f TB/s; per step (2 encoders) %.2f ms" % (tot_t, tot_b / tot_t / 1e6, 2 * tot_t / 1e3))
