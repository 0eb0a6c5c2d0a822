import os, sys, torch, torch.nn as nn, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_common as gc
from deeplio_amd import ops, functional as Fh
from test_gpu_modules import _OraCBR, CBR_CASES
dev = torch.device("cuda:0")
rel = lambda a, b: float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max()) / max(float(b.abs().max()), 1e-30)
N, cin, cout, k, s, p, H, W, pre, bias = CBR_CASES[0]
x = torch.randn(N, cin, H, W, generator=torch.Generator().manual_seed(2))
m = _OraCBR(cin, cout, k, s, p, pre, bias); gc.fill_state(m, 77)
w, b, gam, bet = m.conv.weight.detach().clone(), m.conv.bias.detach().clone(), m.bn.weight.detach().clone(), m.bn.bias.detach().clone()
md = m.double().train()
xi = x.double().requires_grad_(True)
raw_o = md.conv(xi); raw_o.retain_grad()
y_o = md.bn(F.relu(raw_o))
g = torch.randn(y_o.shape, generator=torch.Generator().manual_seed(5))
y_o.backward(g.double())
# hip pieces
xd = x.to(dev)
d = ops.conv_desc(N, cin, H, W, cout, 3, 3, 1, 1, 1, 1)
raw = torch.empty(N, cout, H, W, device=dev)
ops.conv2d_fwd(xd, ops.conv2d_prep_weight(w.to(dev), 0), b.to(dev), raw, d)
print("raw", rel(raw, raw_o))
st = ops.chan_stats(raw, N, cout, 0, cout, H * W, True)
prm = ops.bn_finalize(st, N * H * W, gam.to(dev), 1e-5, 0.1, None, None)
out = torch.empty_like(raw)
ops.bn_apply(raw, cout, 0, prm, bet.to(dev), out, cout, 0, N, cout, H * W, True, False)
print("out", rel(out, y_o))
draw = torch.empty_like(raw)
dg, db = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
ops.bn_bwd(g.to(dev), cout, 0, raw, cout, 0, prm, bet.to(dev), draw, cout, 0, N, cout, H * W, True, False, True, dg, db)
print("draw", rel(draw, raw_o.grad), "dgamma", rel(dg, md.bn.weight.grad), "dbeta", rel(db, md.bn.bias.grad))
# masked diff stats
diff = (draw.double().cpu() - raw_o.grad).abs()
print("n bad", int((diff > 1e-4 * raw_o.grad.abs().max()).sum()), "of", diff.numel())
idx = torch.nonzero(diff > 1e-4 * raw_o.grad.abs().max())[:5]
for i in idx:
    i = tuple(i.tolist()); print(i, float(raw.cpu()[i]), float(raw_o[i]), float(draw.cpu()[i]), float(raw_o.grad[i]))
