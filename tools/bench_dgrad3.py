#!/usr/bin/env python3
"""3x3 data gradient of the Fire expand layers at the headline launch sizes (N = 16), production routing; run once with
DLIO_BX3_PC=0 (conv3x3_bx3_alds_kernel) and once with 1 (conv3x3_bx3_pc_kernel); the last column: the two-piece fp16 kernel
(dlio_conv3x3_h2_fwd) where the launch size has it.  usage: python tools/bench_dgrad3.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from deeplio_amd import ops
from bench_fire import timeit, LAYERS, N
import torch.nn.functional as F
dev = torch.device("cuda:0")
out = []
for name, S, E, H, W in LAYERS:
    d3 = torch.randn(N, E, H, W, device=dev)
    w3 = torch.randn(E, S, 3, 3, device=dev) / (S * 9) ** 0.5
    res = torch.randn(N, S, H, W, device=dev)
    wt3 = ops.conv3x3_bx3_prep(w3, 1)
    dx = torch.empty(N, S, H, W, device=dev)
    g3 = ops.conv_desc(N, E, H, W, S, 3, 3, 1, 1, 1, 1, OH=H, OW=W, res_ctot=S)
    t3 = timeit(lambda: ops.conv3x3_bx3_fwd(d3, wt3, None, dx, g3, residual=res))
    ref = F.conv_transpose2d(d3[:2].double(), w3.double(), padding=1) + res[:2].double()
    err = float((dx[:2].double() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * N * H * W * E * S * 9
    th = float("nan")
    if ops.conv3x3_h2_ok(g3):
        am, wh = d3.abs().max().reshape(1), ops.conv_h2_prepped(w3, 1)
        th = timeit(lambda: ops.conv3x3_h2_fwd(d3, am, wh, None, dx, g3, residual=res))
    out.append("%-6s %4d->%-3d %8.1f us %6.0f TF/s  err %.1e   two-piece %8.1f us" % (name, E, S, t3, fl / t3 / 1e6, err, th))
print("\n".join(out))
