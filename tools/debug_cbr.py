import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max()) / max(float(b.abs().max()), 1e-30)
g = torch.Generator().manual_seed(0)
for shape in [(4, 40, 33, 33), (4, 40, 17, 17), (4, 512, 33, 33), (4, 40, 32, 34)]:
    for pre, post in [(1, 0), (0, 1)]:
        N, C, H, W = shape; HW = H * W
        x = torch.randn(shape, generator=g); dy = torch.randn(shape, generator=g)
        gam, bet = torch.rand(C, generator=g) + .5, torch.randn(C, generator=g)
        xr = x.double().requires_grad_(True)
        xin = F.relu(xr) if pre else xr
        y = F.batch_norm(xin, None, None, gam.double(), bet.double(), True, 0.1, 1e-5)
        if post: y = F.relu(y)
        y.backward(dy.double())
        xd = x.to(dev)
        st = ops.chan_stats(xd, N, C, 0, C, HW, pre)
        prm = ops.bn_finalize(st, N * HW, gam.to(dev), 1e-5, 0.1, None, None)
        yd = torch.empty_like(xd)
        ops.bn_apply(xd, C, 0, prm, bet.to(dev), yd, C, 0, N, C, HW, pre, post)
        dx = torch.empty_like(xd)
        ops.bn_bwd(dy.to(dev), C, 0, xd, C, 0, prm, bet.to(dev), dx, C, 0, N, C, HW, pre, post, True)
        cs = ops.chan_sum(dx, N, C, 0, C, HW)
        print(shape, pre, post, "y %.1e dx %.1e chansum %.1e" % (rel(yd, y), rel(dx, xr.grad), rel(cs, xr.grad.sum((0, 2, 3)))))
# conv pieces at 33x33
N, Cin, Cout, H, W = 4, 256, 512, 33, 33
x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) / 48; dy = torch.randn(N, Cout, H, W, generator=g)
xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
F.conv2d(xr, wr, None, padding=1).backward(dy.double())
d = ops.conv_desc(N, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1)
dw = torch.empty_like(w, device=dev); ops.conv2d_wgrad(x.to(dev), dy.to(dev), dw, d)
dx = torch.empty_like(x, device=dev)
dd = ops.conv_desc(N, Cout, H, W, Cin, 3, 3, 1, 1, 1, 1)
ops.conv2d_fwd(dy.to(dev), ops.conv2d_prep_weight(w.to(dev), 1), None, dx, dd)
print("conv 33x33 256->512: dw %.1e dx %.1e" % (rel(dw, wr.grad), rel(dx, xr.grad)))
