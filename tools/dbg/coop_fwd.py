import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(53)
N, C, C1, H, W = 4, 40, 17, 64, 128
HW = H * W
x = (torch.randn(N, C, H, W, generator=g) * 1.7 + 0.3).to(dev)
gam, bet = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
rm0, rv0 = torch.randn(C, generator=g).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)
x64 = x.double().cpu()
rm64, rv64 = rm0.double().cpu().clone(), rv0.double().cpu().clone()
F.batch_norm(x64, rm64, rv64, gam.double().cpu(), bet.double().cpu(), True, 0.1, 1e-5)
for mode in (0, 1, 1, 0):
    ops.bn_coop_set_mode(mode)
    rm, rv = rm0.clone(), rv0.clone()
    prm = torch.empty(3, C, device=dev); y = torch.empty_like(x)
    ops.bn_coop_fwd(x, C, 0, N, C, C1, HW, (gam[:C1], bet[:C1], rm[:C1], rv[:C1]), (gam[C1:], bet[C1:], rm[C1:], rv[C1:]), 1e-5, 0.1, prm, y, C, 0, True)
    torch.cuda.synchronize()
    e1 = (rm.double().cpu() - rm64).abs(); e2 = (rv.double().cpu() - rv64).abs()
    print("mode", mode, "rm err", float(e1.max()), int(e1.argmax()), "rv err", float(e2.max()), int(e2.argmax()), "errors", ops.bn_coop_errors())
    print("   mean err", float((prm[0].double().cpu() - x64.mean((0, 2, 3))).abs().max()))
