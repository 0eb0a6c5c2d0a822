#!/usr/bin/env python3
"""split-bf16 1x1 convolution against the fp32-MFMA 1x1 kernels on the PointSeg shapes (forward and
data-gradient direction), N = 16 images: error vs fp64 on a small case, then time and TB/s."""
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
for (N, Cin, Cout, H, W) in [(2, 24, 40, 8, 36), (1, 70, 100, 4, 128), (2, 300, 16, 6, 20), (1, 3, 5, 2, 2)]:
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g); res = torch.randn(N, Cout, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double()) + res.double()
    d = ops.conv_desc(N, Cin, H, W, Cout, 1, 1, 1, 1, 0, 0, res_ctot=Cout)
    y = torch.empty(N, Cout, H, W, device=dev); y32 = torch.empty_like(y)
    ops.conv1x1_bx3_fwd(x.to(dev), ops.conv1x1_bx3_prep(w.to(dev), 0), b.to(dev), y, d, residual=res.to(dev))
    ops.conv2d_fwd(x.to(dev), ops.conv2d_prep_weight(w.to(dev), 0), b.to(dev), y32, d, residual=res.to(dev))
    dy = torch.randn(N, Cout, H, W, generator=g)
    xr = x.double().requires_grad_(True)
    F.conv2d(xr, w.double()).backward(dy.double())
    gd = ops.conv_desc(N, Cout, H, W, Cin, 1, 1, 1, 1, 0, 0)
    dx = torch.empty(N, Cin, H, W, device=dev)
    ops.conv1x1_bx3_fwd(dy.to(dev), ops.conv1x1_bx3_prep(w.to(dev), 1), None, dx, gd)
    print("N%d %3d->%3d %dx%d: fwd err bx3 %.2e (fp32 MFMA %.2e) | dgrad err bx3 %.2e" % (
        N, Cin, Cout, H, W, rel(y, ref), rel(y32, ref), rel(dx, xr.grad)))

N = 16
print("%-18s %9s %6s | %9s %6s | %5s" % ("layer", "fp32 us", "TB/s", "bx3 us", "TB/s", "x"))
for name, ci, co, H, W in [("b1.sq 64->16", 64, 16, 64, 512), ("b1.sq 128->16", 128, 16, 64, 512), ("b1.e1 16->64", 16, 64, 64, 512),
                           ("b1 dsq 16->128", 16, 128, 64, 512), ("b2.sq 256->32", 256, 32, 64, 256), ("b2.e1 32->128", 32, 128, 64, 256),
                           ("b2 de1 128->32", 128, 32, 64, 256), ("b3.sq 384->48", 384, 48, 64, 128), ("b3.e1 48->192", 48, 192, 64, 128),
                           ("b3.sq 512->64", 512, 64, 64, 128), ("b3.e1 64->256", 64, 256, 64, 128), ("b3 de1 256->64", 256, 64, 64, 128),
                           ("b3 dsq 64->512", 64, 512, 64, 128), ("b4.e1 64->256", 64, 256, 32, 64), ("b5.sq 768->80", 768, 80, 16, 32),
                           ("b5.e1 80->384", 80, 384, 16, 32)]:
    x = torch.randn(N, ci, H, W, device=dev); w = torch.randn(co, ci, 1, 1, device=dev) * 0.05
    y = torch.empty(N, co, H, W, device=dev)
    d = ops.conv_desc(N, ci, H, W, co, 1, 1, 1, 1, 0, 0)
    w32, wb = ops.conv2d_prep_weight(w, 0), ops.conv1x1_bx3_prep(w, 0)
    t32 = timeit(lambda: ops.conv2d_fwd(x, w32, None, y, d))
    tb = timeit(lambda: ops.conv1x1_bx3_fwd(x, wb, None, y, d))
    byt = 4.0 * N * H * W * (ci + co)
    print("%-18s %9.1f %6.2f | %9.1f %6.2f | %5.2f" % (name, t32, byt / t32 / 1e6, tb, byt / tb / 1e6, t32 / tb))
