#!/usr/bin/env python3
"""split-bf16 3x3 convolution (conv_bx3.hip) against the fp32-MFMA kernel: error vs fp64 torch on a
small case, then time and fp32-equivalent TFLOP/s on the PointSeg expand3x3 shapes (forward and
data-gradient direction), N = 16 images."""
import os, sys, torch, torch.nn.functional as F
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


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.abs().max())


g = torch.Generator().manual_seed(3)
for (N, Cin, Cout, H, W) in [(2, 24, 40, 9, 37), (1, 16, 64, 8, 70), (2, 70, 16, 5, 33), (1, 3, 5, 4, 4)]:
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g); res = torch.randn(N, Cout, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1) + res.double()
    d = ops.conv_desc(N, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1, res_ctot=Cout)
    y = torch.empty(N, Cout, H, W, device=dev)
    ops.conv3x3_bx3_fwd(x.to(dev), ops.conv3x3_bx3_prep(w.to(dev), 0), b.to(dev), y, d, residual=res.to(dev))
    y32 = torch.empty(N, Cout, H, W, device=dev)
    ops.conv2d_fwd(x.to(dev), ops.conv2d_prep_weight(w.to(dev), 0), b.to(dev), y32, d, residual=res.to(dev))
    # data-gradient direction: dx = conv(dy, w^T reversed), pad 1
    dy = torch.randn(N, Cout, H, W, generator=g)
    xr = x.double().requires_grad_(True)
    F.conv2d(xr, w.double(), None, 1, 1).backward(dy.double())
    gd = ops.conv_desc(N, Cout, H, W, Cin, 3, 3, 1, 1, 1, 1)
    dx = torch.empty(N, Cin, H, W, device=dev)
    ops.conv3x3_bx3_fwd(dy.to(dev), ops.conv3x3_bx3_prep(w.to(dev), 1), None, dx, gd)
    print("N%d %3d->%3d %dx%d: fwd err bx3 %.2e (fp32 MFMA %.2e) | dgrad err bx3 %.2e" % (
        N, Cin, Cout, H, W, rel(y, ref), rel(y32, ref), rel(dx, xr.grad)))

N = 16
print("%-22s %9s %7s | %9s %7s | %5s" % ("layer", "fp32 us", "TF/s", "bx3 us", "TF/s eq", "x"))
for name, ci, co, H, W in [("b1.e3 fwd", 16, 64, 64, 512), ("b1.e3 dgrad", 64, 16, 64, 512), ("b2.e3 fwd", 32, 128, 64, 256),
                           ("b2.e3 dgrad", 128, 32, 64, 256), ("b3.0.e3 fwd", 48, 192, 64, 128), ("b3.0.e3 dgrad", 192, 48, 64, 128),
                           ("b3.2.e3 fwd", 64, 256, 64, 128), ("b3.2.e3 dgrad", 256, 64, 64, 128), ("b4.e3 fwd", 64, 256, 32, 64),
                           ("b4.e3 dgrad", 256, 64, 32, 64), ("b5.e3 fwd", 80, 384, 16, 32), ("b5.e3 dgrad", 384, 80, 16, 32),
                           ("flownet conv3_1", 256, 256, 64, 256)]:
    x = torch.randn(N, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    y = torch.empty(N, co, H, W, device=dev)
    d = ops.conv_desc(N, ci, H, W, co, 3, 3, 1, 1, 1, 1)
    w32, wb = ops.conv2d_prep_weight(w, 0), ops.conv3x3_bx3_prep(w, 0)
    t32 = timeit(lambda: ops.conv2d_fwd(x, w32, None, y, d))
    tb = timeit(lambda: ops.conv3x3_bx3_fwd(x, wb, None, y, d))
    fl = 2.0 * N * H * W * ci * co * 9
    print("%-22s %9.1f %7.1f | %9.1f %7.1f | %5.2f" % (name, t32, fl / t32 / 1e6, tb, fl / tb / 1e6, t32 / tb))
