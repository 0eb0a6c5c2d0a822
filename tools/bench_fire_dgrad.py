#!/usr/bin/env python3
"""Fire expand pair, data gradient, at the headline launch sizes (N = 16): expand1x1 and expand3x3 data gradients as two
launches (the second accumulating) against dlio_fire_expand_dgrad; us per launch.  usage: python tools/bench_fire_dgrad.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
from bench_fire import timeit, LAYERS, N
dev = torch.device("cuda:0")
print("%-6s %9s %9s %9s | %9s" % ("layer", "e1 us", "e3 us", "sum", "fused us"))
for name, S, E, H, W in LAYERS:
    d1, d3 = torch.randn(N, E, H, W, device=dev), torch.randn(N, E, H, W, device=dev)
    w3 = torch.randn(E, S, 3, 3, device=dev) / (S * 9) ** 0.5
    w1 = torch.randn(E, S, 1, 1, device=dev) / S ** 0.5
    wt3, wt1 = ops.conv3x3_bx3_prep(w3, 1), ops.conv1x1_bx3_prep(w1, 1)
    dx, dx2 = torch.empty(N, S, H, W, device=dev), torch.empty(N, S, H, W, device=dev)
    g1 = ops.conv_desc(N, E, H, W, S, 1, 1, 1, 1, 0, 0)
    g3 = ops.conv_desc(N, E, H, W, S, 3, 3, 1, 1, 1, 1, OH=H, OW=W, res_ctot=S)
    gf = ops.conv_desc(N, E, H, W, S, 3, 3, 1, 1, 1, 1, OH=H, OW=W)
    t1 = timeit(lambda: ops.conv1x1_bx3_fwd(d1, wt1, None, dx, g1))
    t3 = timeit(lambda: ops.conv3x3_bx3_fwd(d3, wt3, None, dx, g3, residual=dx))
    tf = timeit(lambda: ops.fire_expand_dgrad(d3, wt3, d1, wt1, dx2, gf))
    print("%-6s %9.1f %9.1f %9.1f | %9.1f" % (name, t1, t3, t1 + t3, tf))
