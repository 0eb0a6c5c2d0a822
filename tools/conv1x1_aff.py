#!/usr/bin/env python3
"""Isolated timing of the in-affine (apply-on-load) 1x1 squeeze convolutions of a PointSeg step next to the same
layer without the affine; routing knobs (DLIO_1X1_V4, DLIO_1X1_SPLITK, DLIO_1X1_NR) are read once per process."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
N = 16


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


# (name, Cin, squeeze, H, W): the Fire blocks whose input arrives as (raw, aff)
layers = [("b1.1", 128, 16, 64, 512), ("b2.1", 256, 32, 64, 256), ("b3.1", 384, 48, 64, 128), ("b3.2", 384, 64, 64, 128),
          ("b3.3", 512, 64, 64, 128), ("b4.1", 512, 64, 32, 64), ("b5.1", 768, 80, 16, 32)]
print("%-6s %5s %5s %8s %9s %9s %7s" % ("layer", "cin", "cout", "pixels", "plain us", "aff us", "TB/s"))
for nm, cin, cout, H, W in layers:
    x = torch.randn(N, cin, H, W, device=dev)
    y = torch.empty(N, cout, H, W, device=dev)
    y2 = torch.empty_like(y)
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    aff = torch.randn(3, cin, device=dev); aff[1].abs_()
    wt = ops.conv2d_prep_weight(w, 0)
    d0 = ops.conv_desc(N, cin, H, W, cout, 1, 1, 1, 1, 0, 0)
    d1 = ops.conv_desc(N, cin, H, W, cout, 1, 1, 1, 1, 0, 0, in_relu=1)
    t0 = timeit(lambda: ops.conv2d_fwd(x, wt, None, y, d0))
    t1 = timeit(lambda: ops.conv2d_fwd(x, wt, None, y2, d1, in_aff=(aff[0], aff[1], aff[2])))
    wb = ops.conv1x1_bx3_prep(w, 0)
    y3 = torch.empty_like(y)
    t2 = timeit(lambda: ops.conv1x1_bx3_fwd(x, wb, None, y3, d0))
    t3 = timeit(lambda: ops.conv1x1_bx3_fwd(x, wb, None, y3, d1, in_aff=(aff[0], aff[1], aff[2])))
    ref = torch.nn.functional.conv2d(torch.relu((x - aff[0].view(1, -1, 1, 1)) * aff[1].view(1, -1, 1, 1) + aff[2].view(1, -1, 1, 1)), w)
    err = ((y2 - ref).abs().max() / ref.abs().max()).item()
    byt = 4.0 * N * H * W * (cin + cout)
    err3 = ((y3 - ref).abs().max() / ref.abs().max()).item()
    print("%-6s %5d %5d %8d %9.1f %9.1f %7.2f  err %.1e | bx3 %7.1f aff %7.1f %5.2f TB/s err %.1e" % (
        nm, cin, cout, N * H * W, t0, t1, byt / t1 / 1e6, err, t2, t3, byt / t3 / 1e6, err3))
