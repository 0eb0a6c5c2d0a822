#!/usr/bin/env python3
"""BatchNorm backward of the SQUEEZE layers (16-64 channels on the large planes: 8-34 MB per operand) -- the cooperative
one-launch kernel against the two launches of bn.hip (reduce, apply), whose second read comes out of the L2 / Infinity Cache
at these sizes.  usage: python tools/bench_bn_squeeze.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
N = 16
LAYERS = [("blk1", 16, 64, 512), ("blk2", 32, 64, 256), ("blk3a", 48, 64, 128), ("blk3b", 64, 64, 128)]


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print("%-6s %4s | %9s %9s %9s | MB per operand" % ("layer", "C", "two-launch", "coop", "coop256"))
for name, C, H, W in LAYERS:
    HW = H * W
    raw, dy, dx = (torch.randn(N, C, H, W, device=dev) for _ in range(3))
    g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    prm = torch.empty(3, C, device=dev)
    y = torch.empty_like(raw)
    ops.bn_coop_fwd(raw, C, 0, N, C, C, HW, (g, b, None, None), None, 1e-5, 0.1, prm, y, C, 0, True)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    # a writer of the gradient in front of each launch, as in the step (the gradient is fresh in the caches)
    def two():
        ops.bn_bwd_fused(dy, C, 0, raw, C, 0, (prm[0], prm[1], prm[2]), b, dx, C, 0, N, C, HW, False, True, True, dg, db)
    def coop():
        ops.bn_coop_bwd(dy, C, 0, raw, C, 0, prm, b, None, dx, None, dg, db, None, None, False, N, C, C, HW, True)
    t2 = timeit(two)
    tc = timeit(coop)
    ops.bn_coop_set_cus(256)
    tc2 = timeit(coop)
    ops.bn_coop_set_cus(0)
    print("%-6s %4d | %9.1f %9.1f %9.1f | %.1f" % (name, C, t2, tc, tc2, 4e-6 * N * C * HW))
print("coop errors:", ops.bn_coop_errors())
