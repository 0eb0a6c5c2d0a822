import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeplio_amd import ops
dev = torch.device("cuda:0")
N, Cin, Cout, H, W = 16, 10, 64, 64, 2048
x = torch.randn(N, Cin, H, W, device=dev); w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 5, device=dev) * 0.05)
d = ops.conv_desc(N, Cin, H, W, Cout, 3, 5, 1, 2, 1, 2)
y = torch.empty(N, Cout, d.OH, d.OW, device=dev)
wb = ops.conv_bx3_prepped(w, 0); w32 = ops.conv2d_prep_weight(w.detach(), 0)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
print("fp32 %.1f us   bx3 %.1f us" % (t(lambda: ops.conv2d_fwd(x, w32, None, y, d)), t(lambda: ops.conv3x5s2_bx3_fwd(x, wb, None, y, d))))
