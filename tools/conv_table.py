#!/usr/bin/env python3
"""Isolated per-layer timing of every PointSeg convolution at the headline shape (N=16 images per
encoder): forward, data gradient, weight gradient.  Prints us, TFLOP/s, TB/s (min traffic) and the
share of the per-step conv time (x2 encoders)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
from deeplio_amd import functional as Fh
dev = torch.device("cuda:0")
N = 16


def timeit(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


layers = [("stem", 10, 64, 3, 5, 1, 2, 64, 2048)]
H, W = 64, 512
blocks = [("b1", [(64, 16, 64), (128, 16, 64)], (1, 2)), ("b2", [(128, 32, 128), (256, 32, 128)], (1, 2)),
          ("b3", [(256, 48, 192), (384, 48, 192), (384, 64, 256), (512, 64, 256)], (2, 2)),
          ("b4", [(512, 64, 256), (512, 64, 256)], (2, 2)), ("b5", [(512, 80, 384), (768, 80, 384)], None)]
for bn, fires, pool in blocks:
    for i, (ci, sq, e) in enumerate(fires):
        layers += [("%s.%d.sq" % (bn, i), ci, sq, 1, 1, 1, 1, H, W), ("%s.%d.e1" % (bn, i), sq, e, 1, 1, 1, 1, H, W),
                   ("%s.%d.e3" % (bn, i), sq, e, 3, 3, 1, 1, H, W)]
    if pool:
        H, W = H // pool[0], W // pool[1]
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
rows = []
for name, ci, co, kh, kw, sh, sw, H, W in layers:
    ph, pw = kh // 2, kw // 2
    OH, OW = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    x = torch.randn(N, ci, H, W, device=dev); w = torch.randn(co, ci, kh, kw, device=dev) * 0.05
    y = torch.randn(N, co, OH, OW, device=dev)
    d = ops.conv_desc(N, ci, H, W, co, kh, kw, sh, sw, ph, pw)
    fl = 2.0 * N * OH * OW * ci * co * kh * kw
    byt = 4.0 * N * (H * W * ci + OH * OW * co)
    wt = ops.conv2d_prep_weight(w, 0)
    t_f = timeit(lambda: ops.conv2d_fwd(x, wt, None, y, d))
    dx = torch.empty_like(x)
    t_d = timeit(lambda: Fh.conv_dgrad(y, w, d, dx, ci, 0)) if name != "stem" else 0.0
    dw = torch.empty_like(w)
    t_w = timeit(lambda: ops.conv2d_wgrad(x, y, dw, d))
    rows.append((name, ci, co, kh, kw, OH, OW, fl, byt, t_f, t_d, t_w))
    tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w
T = sum(tot.values())
print("%-9s %4s %4s %3s %9s | %8s %6s %5s | %8s %6s %5s | %8s %6s %5s | %5s" % (
    "layer", "ci", "co", "k", "out", "fwd us", "TF/s", "TB/s", "dgrad us", "TF/s", "TB/s", "wgrad us", "TF/s", "TB/s", "share"))
for name, ci, co, kh, kw, OH, OW, fl, byt, t_f, t_d, t_w in rows:
    f = lambda t: (fl / t / 1e6, byt / t / 1e6) if t > 0 else (0.0, 0.0)
    print("%-9s %4d %4d %dx%d %4dx%-4d | %8.1f %6.1f %5.2f | %8.1f %6.1f %5.2f | %8.1f %6.1f %5.2f | %4.1f%%" % (
        name, ci, co, kh, kw, OH, OW, t_f, *f(t_f), t_d, *f(t_d), t_w, *f(t_w), 100 * (t_f + t_d + t_w) / T))
print("totals per encoder (us):", {k: round(v, 1) for k, v in tot.items()}, " x2 encoders = %.2f ms/step" % (2 * T / 1e3))
