#!/usr/bin/env python3
"""Fire expand pair at the headline launch sizes (N = 16): the separate path (BatchNorm apply, expand1x1 and expand3x3 on
conv1x1_bx3 / conv3x3_bx3_alds) against the fused one (dlio_bn_split16 + dlio_fire_expand_fwd); us per launch by hipEvents.
usage: python tools/bench_fire.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
LAYERS = [("blk1", 16, 64, 64, 512), ("blk2", 32, 128, 64, 256), ("blk3a", 48, 192, 64, 128), ("blk3b", 64, 256, 64, 128),
          ("blk4", 64, 256, 32, 64), ("blk5", 80, 384, 16, 32)]
N = 16


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print("%-6s %9s %9s %9s | %9s %9s | %9s %9s" % ("layer", "bn us", "e1 us", "e3 us", "split us", "fused us", "old sum", "new sum"))
for name, S, E, H, W in LAYERS:
    raw = torch.randn(N, S, H, W, device=dev)
    gamma, beta = torch.rand(S, device=dev) + 0.5, torch.randn(S, device=dev)
    rm, rv = torch.zeros(S, device=dev), torch.ones(S, device=dev)
    w3 = torch.randn(E, S, 3, 3, device=dev) / (S * 9) ** 0.5
    w1 = torch.randn(E, S, 1, 1, device=dev) / S ** 0.5
    b3, b1 = torch.randn(E, device=dev), torch.randn(E, device=dev)
    w3t, w1t = ops.conv3x3_bx3_prep(w3, 0), ops.conv1x1_bx3_prep(w1, 0)
    act = torch.empty(N, S, H, W, device=dev)
    y = torch.empty(N, 2 * E, H, W, device=dev)
    y2 = torch.empty(N, 2 * E, H, W, device=dev)
    planes = ops.fire_planes(N, S, H, W, dev)
    d1 = ops.conv_desc(N, S, H, W, E, 1, 1, 1, 1, 0, 0, out_ctot=2 * E, out_coff=0)
    d3 = ops.conv_desc(N, S, H, W, E, 3, 3, 1, 1, 1, 1, out_ctot=2 * E, out_coff=E)
    t_bn = timeit(lambda: ops.bn_train_apply(raw, S, 0, gamma, beta, 1e-5, 0.1, rm, rv, act, S, 0, N, S, H * W, False, True))
    t_e1 = timeit(lambda: ops.conv1x1_bx3_fwd(act, w1t, b1, y, d1))
    t_e3 = timeit(lambda: ops.conv3x3_bx3_fwd(act, w3t, b3, y, d3))
    t_sp = timeit(lambda: ops.bn_split16(raw, S, 0, gamma, beta, 1e-5, 0.1, rm, rv, act, S, 0, planes, N, S, H, W, True))
    t_fu = timeit(lambda: ops.fire_expand_fwd(planes, w3t, w1t, b3, b1, y2, N, S, H, W, E, 2 * E, 0))
    ok = torch.equal(y, y2)
    fl = 2.0 * N * H * W * E * S * 10
    print("%-6s %9.1f %9.1f %9.1f | %9.1f %9.1f | %9.1f %9.1f  %s  fused %.0f TF/s, %.2f TB/s out" % (
        name, t_bn, t_e1, t_e3, t_sp, t_fu, t_bn + t_e1 + t_e3, t_sp + t_fu, "bit-equal" if ok else "DIFFERENT",
        fl / t_fu / 1e6, 4.0 * N * H * W * 2 * E / t_fu / 1e6))
