#!/usr/bin/env python3
"""Forward BatchNorm of a Fire block's concat buffer, N = 16: the cooperative one-launch kernel against the two phases of
bn.hip's path timed separately (phase 1 = statistics partials, phase 2 = finalise + apply from the partials) -- what an
apply pass would cost if the statistics came out of the convolution's epilogue.  usage: python tools/bench_bn_phases.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
from deeplio_amd.ops import lib, _ptr, _stream, check, _stats_ws
dev = torch.device("cuda:0")
LAYERS = [("blk1", 64, 64, 512), ("blk2", 128, 64, 256), ("blk3a", 192, 64, 128), ("blk3b", 256, 64, 128)]
N = 16


def timeit(fn, reps=20):
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


print("%-6s | %8s %8s | %8s %8s %8s | %8s" % ("layer", "coop", "coop+gap", "phase1", "phase2", "ph2+res", "stats"))
for name, E, H, W in LAYERS:
    CE, HW = 2 * E, H * W
    raw = torch.randn(N, CE, H, W, device=dev); res = torch.randn(N, CE, H, W, device=dev); out = torch.empty(N, CE, H, W, device=dev)
    g, b = torch.rand(CE, device=dev) + 0.5, torch.randn(CE, device=dev)
    rm, rv = torch.zeros(CE, device=dev), torch.ones(CE, device=dev)
    s1 = (g[:E], b[:E], rm[:E], rv[:E]); s2 = (g[E:], b[E:], rm[E:], rv[E:])
    prm = torch.empty(3, CE, device=dev); gap = torch.empty(N, CE, device=dev)
    ws = _stats_ws(N, CE, HW, dev)

    def phase(ph, r=None):
        check(lib.dlio_bn_train_apply(_ptr(raw), N, CE, 0, CE, HW, 0, 1, _ptr(g), _ptr(b), 1e-5, 0.1, _ptr(rm), _ptr(rv), _ptr(prm[0]),
                                      _ptr(prm[1]), _ptr(prm[2]), _ptr(r), CE, 0, _ptr(out), CE, 0, None, 0, 0, _ptr(ws), ws.numel(),
                                      ph, 1.0, None, None, None, None, _stream()), "x")
    t_c = timeit(lambda: ops.bn_coop_fwd(raw, CE, 0, N, CE, E, HW, s1, s2, 1e-5, 0.1, prm, out, CE, 0, True))
    t_g = timeit(lambda: ops.bn_coop_fwd(raw, CE, 0, N, CE, E, HW, s1, s2, 1e-5, 0.1, prm, out, CE, 0, True, gap_out=gap, gap_ctot=CE))
    t_1 = timeit(lambda: phase(1))
    phase(1)
    t_2 = timeit(lambda: phase(2))
    t_2r = timeit(lambda: phase(2, res))
    t_s = timeit(lambda: ops.bn_train_stats(raw, N, CE, 0, CE, HW, False, g, 1e-5, 0.1, rm, rv))
    print("%-6s | %8.1f %8.1f | %8.1f %8.1f %8.1f | %8.1f" % (name, t_c, t_g, t_1, t_2, t_2r, t_s))
