"""conditioning of Fire -> GAP gradients (the last block of PSEncoder): fp32 path with inputs perturbed by
bf16-sized noise vs the bf16 path"""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_common as gc
from deeplio_amd import nets
from deeplio_amd import functional as Fh
dev = torch.device("cuda:0")


def cmp(a, b):
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    return float(F.cosine_similarity(a, b, dim=0)), float((a - b).norm() / b.norm().clamp_min(1e-30))


def run(N, H, W, mode, x, dfeat):
    f = nets.Fire(768, 80, 384, 384, bypass=None)
    gc.fill_state(f, 77); f.to(dev).train()
    if mode == "bf16":
        xa = x.to(dev).bfloat16().requires_grad_(True)
    elif mode == "fp32":
        xa = x.to(dev).requires_grad_(True)
    else:   # fp32 arithmetic on an input perturbed by bf16-sized relative noise
        noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(9)) * 2 ** -9
        xa = (x * (1 + noise)).to(dev).requires_grad_(True)
    y = f(xa)
    feat = nets._gap(y)
    feat.backward(dfeat.to(dev))
    return feat, xa.grad.float(), {k: p.grad.clone() for k, p in f.named_parameters()}


for N, H, W in ((4, 4, 16), (16, 16, 32)):
    x = F.relu(torch.randn(N, 768, H, W, generator=torch.Generator().manual_seed(1))).bfloat16().float()
    dfeat = torch.randn(N, 768, generator=torch.Generator().manual_seed(2))
    ref = run(N, H, W, "fp32", x, dfeat)
    for mode in ("bf16", "noisy"):
        got = run(N, H, W, mode, x, dfeat)
        print(N, H, W, mode, "feat", cmp(got[0], ref[0]), "dx", cmp(got[1], ref[1]))
        for k in ("squeeze.weight", "expand1x1.weight", "expand3x3.weight", "expand3x3_bn.weight", "expand3x3_bn.bias"):
            print("    %-22s cos %.4f relL2 %.3e" % ((k,) + cmp(got[2][k], ref[2][k])))
