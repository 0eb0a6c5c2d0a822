#!/usr/bin/env python3
"""3x3 stride-1 weight gradient at the six PointSeg expand3x3 shapes of the headline step (N = 16): us per launch,
fp32-equivalent TFLOP/s, fraction of the split-bf16 ceiling (2516.8 / 6 = 419.5 TF/s), operand bytes / time.
   python tools/bench_wgrad3.py [bf16 | h2]      (h2: the two-piece fp16 split, dlio_conv3x3_wgrad_h2)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
N = 16
bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
h2 = len(sys.argv) > 1 and sys.argv[1] == "h2"
if bf16:
    from deeplio_amd import mixed
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us
shapes = [("blk1", 16, 64, 64, 512), ("blk2", 32, 128, 64, 256), ("blk3a", 48, 192, 64, 128), ("blk3b", 64, 256, 64, 128),
          ("blk4", 64, 256, 32, 64), ("blk5", 80, 384, 16, 32)]
tot = 0.0
for name, ci, co, H, W in shapes:
    dt = torch.bfloat16 if bf16 else torch.float32
    x = torch.randn(N, ci, H, W, device=dev).to(dt); dy = torch.randn(N, co, H, W, device=dev).to(dt)
    dw = torch.empty(co, ci, 3, 3, device=dev)
    d = ops.conv_desc(N, ci, H, W, co, 3, 3, 1, 1, 1, 1)
    fl = 2.0 * N * H * W * ci * co * 9
    by = (2.0 if bf16 else 4.0) * N * H * W * (ci + co)
    if h2:
        ax, ay = x.abs().max().reshape(1), dy.abs().max().reshape(1)
        us = timeit(lambda: ops.conv3x3_wgrad_h2(x, ax, dy, ay, dw, d))
    elif bf16:
        us = timeit(lambda: mixed.conv_wgrad(x, dy, dw, d))
    else:
        us = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, d))
    ceil = 2516.8 if bf16 else 419.5
    tot += 4 * us
    print("%-6s ci=%3d co=%3d %2dx%-3d: %7.1f us  %6.1f TF/s  %4.2f of %s  %5.2f TB/s" % (name, ci, co, H, W, us, fl / us / 1e6,
          fl / us / 1e6 / ceil, ceil, by / us / 1e6))
print("sum over the step's 24 launches: %.2f ms" % (tot / 1e3))
