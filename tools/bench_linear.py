#!/usr/bin/env python3
"""skinny-M nn.Linear kernels at the shapes of the step's serial middle (fusion, odometry bi-LSTM 1024x2, heads):
forward, data gradient, weight gradient -- us per launch and weight-stream GB/s (4 N K bytes per call)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
tot = [0, 0, 0]
for name, M, N, K, cnt in [("lstm W_hh step", 8, 4096, 1024, 8), ("lstm l0 W_ih", 16, 4096, 256, 2), ("lstm l1 W_ih", 16, 4096, 2048, 2),
                           ("fc1 lidar", 16, 128, 768, 1), ("soft fusion", 16, 128, 256, 2), ("heads", 16, 3, 1024, 2)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    dz = torch.randn(M, N, device=dev); dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    tf = timeit(lambda: ops.linear_fwd(x, w, b))
    td = timeit(lambda: ops.linear_bwd_data(dz, w, M))
    tw = timeit(lambda: ops.linear_bwd_weight(dz, x, M, N, K, dw=dw, db=db, accumulate=True))
    by = 4.0 * N * K
    print("%-16s M=%2d N=%4d K=%4d | fwd %6.1f us %6.0f GB/s | dgrad %6.1f us %6.0f GB/s | wgrad %6.1f us %6.0f GB/s (r+w)" % (
        name, M, N, K, tf, by / tf / 1e3, td, by / td / 1e3, tw, 2 * by / tw / 1e3))
    tot[0] += cnt * tf; tot[1] += cnt * td; tot[2] += cnt * tw
print("per step (launch counts of the tail): fwd %.0f us, dgrad %.0f us, wgrad %.0f us" % tuple(tot))
