#!/usr/bin/env python3
"""BatchNorm of a Fire block's concat buffer at the headline launch sizes (N = 16): bn.hip's two launches per layer
(statistics partials + plane apply; reductions + plane backward) against the one-launch kernels of bn_small.hip (cooperative
for large planes); us per BatchNorm of the whole buffer, forward (with residual) and backward.  usage: python tools/bench_bn.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
LAYERS = [("blk1", 64, 64, 512), ("blk2", 128, 64, 256), ("blk3a", 192, 64, 128), ("blk3b", 256, 64, 128), ("blk4", 256, 32, 64),
          ("blk5", 384, 16, 32)]
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


print("%-6s | %9s %9s %9s %9s | %9s %9s %9s | GB/s one-launch fwd, stream, bwd, bwd+pool" % ("layer", "fwd 2x2", "fwd one", "stream", "str+pool", "bwd 2x2", "bwd one", "bwd pool"))
for name, E, H, W in LAYERS:
    CE, HW = 2 * E, H * W
    raw = torch.randn(N, CE, H, W, device=dev)
    res = torch.randn(N, CE, H, W, device=dev)
    dy = torch.randn(N, CE, H, W, device=dev)
    out = torch.empty(N, CE, H, W, device=dev)
    g, b = torch.rand(CE, device=dev) + 0.5, torch.randn(CE, device=dev)
    rm, rv = torch.zeros(CE, device=dev), torch.ones(CE, device=dev)
    s1 = (g[:E], b[:E], rm[:E], rv[:E]); s2 = (g[E:], b[E:], rm[E:], rv[E:])
    prm = torch.empty(3, CE, device=dev)
    d1, d3 = torch.empty(N, E, H, W, device=dev), torch.empty(N, E, H, W, device=dev)
    dg, db = torch.empty(CE, device=dev), torch.empty(CE, device=dev)
    coop = ops.bn_coop_ok(N, HW)
    one_f, one_b = (ops.bn_coop_fwd, ops.bn_coop_bwd) if coop else (ops.bn_small_fwd, ops.bn_small_bwd)

    def f_old():
        for off, s in ((0, s1), (E, s2)):
            ops.bn_train_apply(raw, CE, off, s[0], s[1], 1e-5, 0.1, s[2], s[3], out, CE, off, N, E, HW, False, True, res, CE, off)

    def b_old():
        for off, s, dx in ((0, s1, d1), (E, s2, d3)):
            ops.bn_bwd_fused(dy, CE, off, raw, CE, off, (prm[0, off:off + E], prm[1, off:off + E], prm[2, off:off + E]), s[1], dx, E, 0,
                             N, E, HW, False, True, True, dg[off:off + E], db[off:off + E])
    t_fo = timeit(f_old)
    t_fn = timeit(lambda: one_f(raw, CE, 0, N, CE, E, HW, s1, s2, 1e-5, 0.1, prm, out, CE, 0, True, residual=res, r_ctot=CE, r_coff=0))
    t_bo = timeit(b_old)
    t_bn = timeit(lambda: one_b(dy, CE, 0, raw, CE, 0, prm, b[:E], b[E:], d1, d3, dg[:E], db[:E], dg[E:], db[E:], False, N, CE, E, HW, True))
    tb = 4.0 * N * CE * HW
    # round 5: statistics known (expand epilogue) -> streaming apply; the same + the pool behind the SELayer; backward with
    # the pooled gradient routed on load
    aff = torch.stack([torch.zeros(CE, device=dev), g, b])
    t_st = timeit(lambda: ops.bn_aff_apply(raw, CE, 0, N, CE, HW, aff, out, CE, 0, residual=res, r_ctot=CE, r_coff=0))
    SH = 2 if name.startswith("blk3") else 1
    t_sp = t_bp = float("nan")
    if coop and ops.bn_coop_pool_ok(N, H, W, SH):
        t_sp = timeit(lambda: ops.bn_aff_pool_fwd(raw, CE, 0, N, CE, H, W, SH, aff, residual=res, r_ctot=CE, r_coff=0))
        yp, idx = ops.maxpool2d_fwd(out, 3, SH, 2, 1, 1, False)
        dyp = torch.randn_like(yp)
        xs, xa = torch.rand(N, CE, device=dev) + 0.1, torch.randn(N, CE, device=dev) * 1e-3
        t_bp = timeit(lambda: ops.bn_coop_bwd_pool(None, CE, 0, (dyp, idx, xs, xa, SH), raw, CE, 0, prm, b[:E], b[E:], d1, d3, dg[:E],
                                                   db[:E], dg[E:], db[E:], False, N, CE, E, H, W, True))
    pb = tb * 2 + 5.0 * N * CE * HW / (2 * SH)
    print("%-6s | %9.1f %9.1f %9.1f %9.1f | %9.1f %9.1f %9.1f | %6.0f %6.0f %6.0f %6.0f  %s" % (
        name, t_fo, t_fn, t_st, t_sp, t_bo, t_bn, t_bp, 3 * tb / t_fn / 1e3, 3 * tb / t_st / 1e3, 3 * tb / t_bn / 1e3, pb / t_bp / 1e3,
        "coop" if coop else "small"))
print("coop errors:", ops.bn_coop_errors())
