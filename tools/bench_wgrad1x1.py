import os, sys, torch
sys.path.insert(0, "/root/repo")
from deeplio_amd import ops
dev = torch.device("cuda:0"); N = 16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
tot = 0
for name, ci, co, H, W, cnt in [("b1.sq0", 64, 16, 64, 512, 1), ("b1.e1", 16, 64, 64, 512, 2), ("b1.sq1", 128, 16, 64, 512, 1), ("b2.sq0", 128, 32, 64, 256, 1), ("b2.e1", 32, 128, 64, 256, 2),
                                ("b2.sq1", 256, 32, 64, 256, 1), ("b3.sq0", 256, 48, 64, 128, 1), ("b3.e1a", 48, 192, 64, 128, 2), ("b3.sq1", 384, 48, 64, 128, 1), ("b3.sq2", 384, 64, 64, 128, 1),
                                ("b3.e1b", 64, 256, 64, 128, 2), ("b3.sq3", 512, 64, 64, 128, 1), ("b4.sq", 512, 64, 32, 64, 2), ("b4.e1", 64, 256, 32, 64, 2), ("b5.sq0", 512, 80, 16, 32, 1),
                                ("b5.e1", 80, 384, 16, 32, 2), ("b5.sq1", 768, 80, 16, 32, 1)]:
    x = torch.randn(N, ci, H, W, device=dev); dy = torch.randn(N, co, H, W, device=dev); dw = torch.empty(co, ci, 1, 1, device=dev)
    d = ops.conv_desc(N, ci, H, W, co, 1, 1, 1, 1, 0, 0)
    us = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, d)); tot += cnt * us
    print("%-7s %3d->%3d %2dx%-3d %6.1f us %5.2f TB/s" % (name, ci, co, H, W, us, 4.0 * N * H * W * (ci + co) / us / 1e6))
print("per encoder %.1f us -> x2 = %.2f ms/step" % (tot, 2 * tot / 1e3))
