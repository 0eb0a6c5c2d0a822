"""what one fork costs the forking stream: N small launches on stream A, bare / + event record per launch / + a companion
stream waiting for each event and launching there too (the shape of functional._forked)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
a = torch.randn(1 << 16, device=dev); b = torch.randn(1 << 16, device=dev); y = torch.empty_like(a); y2 = torch.empty_like(a)
big = torch.randn(1 << 26, device=dev); bigy = torch.empty_like(big)
A, B = torch.cuda.Stream(), torch.cuda.Stream()
while ops.streams_share_queue(A, B):
    B = torch.cuda.Stream()
N = 400


def run(mode, tensors):
    x0, x1, o, o2 = tensors
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(A):
        e0.record(A)
        for _ in range(N):
            ops.ew_binary(x0, x1, 0, out=o) if "out" in ops.ew_binary.__code__.co_varnames else ops.ew_binary(x0, x1, 0)
            if mode >= 1:
                ev = torch.cuda.Event(); ev.record(A)
            if mode >= 2:
                B.wait_event(ev)
                with torch.cuda.stream(B):
                    ops.ew_binary(x0, x1, 0)
            if mode == 3:
                A.wait_stream(B)
        e1.record(A)
    host = (time.perf_counter() - t0) * 1e6 / N
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N, host


for name, t in (("small (256 KB)", (a, b, y, y2)), ("large (256 MB)", (big, big, bigy, bigy))):
    for mode, what in ((0, "bare"), (1, "+ record"), (2, "+ record, companion waits + launches"), (3, "+ join back every time")):
        run(mode, t)
        g, h = run(mode, t)
        print("%-16s %-40s %7.2f us per launch on A (host %.1f us)" % (name, what, g, h))
