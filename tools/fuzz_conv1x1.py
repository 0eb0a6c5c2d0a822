#!/usr/bin/env python3
"""every routing branch of the 1x1 forward kernels (fp32 float4 / split-K / dword, split-bf16; with and without
the apply-on-load transform, bias, residual) against fp64 torch over a grid of shapes"""
import itertools, os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
bad = n = 0
for N, (H, W), Cin, Cout, aff, res in itertools.product((1, 2, 4, 16), ((4, 16), (8, 32), (16, 64), (32, 128), (64, 128), (6, 10)),
                                                        (16, 24, 128, 256, 384, 512), (16, 48, 64, 80), (False, True), (False, True)):
    if N * H * W * (Cin + Cout) > 3e7:
        continue
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    a = torch.randn(3, Cin, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g) if res else None
    xin = x.double()
    if aff:
        xin = torch.relu((xin - a[0].double().view(1, -1, 1, 1)) * a[1].double().view(1, -1, 1, 1) + a[2].double().view(1, -1, 1, 1))
    ref = F.conv2d(xin, w.double(), b.double())
    if res:
        ref = ref + r.double()
    d = ops.conv_desc(N, Cin, H, W, Cout, 1, 1, 1, 1, 0, 0, in_relu=1 if aff else 0, res_ctot=Cout if res else 0)
    ad = a.to(dev)
    ia = (ad[0], ad[1], ad[2]) if aff else None
    xd, wd, bd, rd = x.to(dev), w.to(dev), b.to(dev), (r.to(dev) if res else None)
    outs = {}
    y = torch.empty(N, Cout, H, W, device=dev)
    ops.conv2d_fwd(xd, ops.conv2d_prep_weight(wd, 0), bd, y, d, in_aff=ia, residual=rd)
    outs["fp32"] = y
    if (H * W) % 4 == 0:
        y2 = torch.empty(N, Cout, H, W, device=dev)
        ops.conv1x1_bx3_fwd(xd, ops.conv1x1_bx3_prep(wd, 0), bd, y2, d, residual=rd, in_aff=ia)
        outs["bx3"] = y2
    for k, v in outs.items():
        n += 1
        e = float((v.double().cpu() - ref).abs().max() / ref.abs().max())
        if not e < 5e-6:
            bad += 1
            if bad < 40:
                print("BAD %-5s N%d %dx%d %d->%d aff=%d res=%d err %.2e" % (k, N, H, W, Cin, Cout, aff, res, e))
print("%d cases, %d bad" % (n, bad))
