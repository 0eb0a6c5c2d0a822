#!/usr/bin/env python3
"""micro-bench of single conv launches (fwd / dgrad / wgrad) at PointSeg headline shapes"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
N = 16
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us
shapes = [("b1.e3 fwd", 16, 64, 3, 64, 512), ("b1.e3 dgrad", 64, 16, 3, 64, 512), ("b2.e3 dgrad", 128, 32, 3, 64, 256),
          ("b3.e3 fwd", 48, 192, 3, 64, 128), ("b3.e3 dgrad", 256, 64, 3, 64, 128), ("b3.e3b fwd", 64, 256, 3, 64, 128),
          ("b4.e3 dgrad", 256, 64, 3, 32, 64), ("b5.e3 dgrad", 384, 80, 3, 16, 32), ("b5.e3 fwd", 80, 384, 3, 16, 32),
          ("b1.sq fwd", 64, 16, 1, 64, 512), ("b3.sq fwd", 512, 64, 1, 64, 128), ("b3.sq dgrad", 64, 512, 1, 64, 128)]
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for name, ci, co, k, H, W in shapes:
    x = torch.randn(N, ci, H, W, device=dev); w = torch.randn(co, ci, k, k, device=dev) * 0.05
    y = torch.empty(N, co, H, W, device=dev)
    d = ops.conv_desc(N, ci, H, W, co, k, k, 1, 1, k // 2, k // 2)
    fl = 2.0 * N * H * W * ci * co * k * k
    if which == "fwd":
        wt = ops.conv2d_prep_weight(w, 0)
        us = timeit(lambda: ops.conv2d_fwd(x, wt, None, y, d))
    else:
        dw = torch.empty_like(w)
        us = timeit(lambda: ops.conv2d_wgrad(x, y, dw, d))
    print("%-12s %s ci=%3d co=%3d k=%d %dx%d: %7.1f us  %6.1f TF/s  %5.2f TB/s" % (name, which, ci, co, k, H, W, us, fl / us / 1e6, 4.0 * N * H * W * (ci + co) / us / 1e6))
