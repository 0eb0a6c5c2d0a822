"""achievable HBM bandwidth on this box: pure write (fill), pure read (sum), copy"""
import torch
dev = torch.device("cuda:0")
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3
for mb in (128, 512, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    tw = t(lambda: x.zero_()); tr = t(lambda: x.sum()); tc = t(lambda: y.copy_(x))
    print("%5d MB: write %.2f TB/s, read %.2f TB/s, copy %.2f TB/s (read+write)" % (mb, n * 4 / tw / 1e12, n * 4 / tr / 1e12, 2 * n * 4 / tc / 1e12))
