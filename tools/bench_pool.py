#!/usr/bin/env python3
"""isolated timing of the 3x3 / stride-(1,2) max-pool kernels at the PointSeg headline shapes"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (N, C, H, W) in [(16, 64, 64, 1024), (16, 128, 64, 512), (16, 256, 64, 256)]:
    x = torch.randn(N, C, H, W, device=dev)
    y, idx = ops.maxpool2d_fwd(x, 3, 1, 2, 1, 1)
    dy = torch.randn_like(y)
    nb = x.numel() * 4 + y.numel() * 5
    t_f = timeit(lambda: ops.maxpool2d_fwd(x, 3, 1, 2, 1, 1))
    t_b = timeit(lambda: ops.maxpool2d_bwd(dy, idx, x.shape, 3, 1, 2, 1, 1))
    t_d = timeit(lambda: ops.maxpool2d_bwd_dot(dy, idx, x, 3, 1, 2, 1, 1)) if hasattr(ops, "maxpool2d_bwd_dot") else 0
    print("%s fwd %.1f us %.2f TB/s | bwd %.1f us %.2f TB/s | bwd_dot %.1f us %.2f TB/s" % (
        (N, C, H, W), t_f, nb / t_f / 1e6, t_b, nb / t_b / 1e6, t_d, nb / max(t_d, 1e-9) / 1e6))
