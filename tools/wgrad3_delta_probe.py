#!/usr/bin/env python3
"""delta-function probes of the 3x3 weight-gradient kernel: which (co, ci, ky, kx) does a single (x, dy) pair hit"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
N, Cin, Cout, H, W = 1, 16, 32, 8, 64
d = ops.conv_desc(N, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1)
def probe(c, r, s, o, r2, s2, xv=1.0, dv=1.0):
    x = torch.zeros(N, Cin, H, W, device=dev); dy = torch.zeros(N, Cout, H, W, device=dev)
    x[0, c, r, s] = xv; dy[0, o, r2, s2] = dv
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=dev)
    ops.conv2d_wgrad(x, dy, dw, d)
    nz = dw.nonzero().tolist()
    print("x[c=%d,r=%d,s=%d] dy[o=%d,r=%d,s=%d] -> expect (o,c,ky=%d,kx=%d)=%g; got %s" % (
        c, r, s, o, r2, s2, r - r2 + 1, s - s2 + 1, xv * dv, [(tuple(i), float(dw[tuple(i)])) for i in nz[:8]]))
probe(3, 4, 10, 5, 4, 10)
probe(3, 4, 10, 5, 4, 9)
probe(3, 4, 10, 5, 4, 11)
probe(3, 5, 10, 5, 4, 10)
probe(3, 3, 10, 5, 4, 10)
probe(3, 4, 16, 5, 4, 15)
probe(3, 4, 15, 5, 4, 16)
probe(3, 4, 31, 5, 4, 32)
probe(3, 4, 32, 5, 4, 31)
probe(3, 4, 10, 5, 4, 10, 1.2345678, 0.87654321)
probe(9, 0, 0, 20, 0, 0, 3.14159, 2.71828)
probe(9, 7, 63, 20, 7, 63, 3.14159, 2.71828)
