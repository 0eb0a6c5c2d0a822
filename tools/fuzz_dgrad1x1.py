#!/usr/bin/env python3
"""data-gradient direction of the 1x1 kernels (mode-1 weight layouts, accumulate = residual aliasing the output, channel
slices) against fp64 torch: split-bf16 and fp32 paths"""
import itertools, os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(2)
bad = n = 0
for N, (H, W), Cf_in, Cf_out, acc in itertools.product((1, 4, 8), ((4, 16), (16, 64), (64, 64), (8, 32)), (16, 64, 128, 256, 384, 512, 768),
                                                       (16, 32, 48, 64, 80, 128, 192), (False, True)):
    if N * H * W * (Cf_in + Cf_out) > 3e7:
        continue
    # forward conv Cf_in -> Cf_out; dgrad: dy [N, Cf_out] -> dx [N, Cf_in]
    w = torch.randn(Cf_out, Cf_in, 1, 1, generator=g) / Cf_in ** 0.5
    dy = torch.randn(N, Cf_out, H, W, generator=g)
    prev = torch.randn(N, Cf_in + 5, H, W, generator=g)
    ref = F.conv_transpose2d(dy.double(), w.double())
    if acc:
        ref = ref + prev[:, 3:3 + Cf_in].double()
    gd = ops.conv_desc(N, Cf_out, H, W, Cf_in, 1, 1, 1, 1, 0, 0, in_ctot=Cf_out, in_coff=0, out_ctot=Cf_in + 5, out_coff=3,
                       res_ctot=(Cf_in + 5) if acc else 0, res_coff=3 if acc else 0)
    for k in ("fp32", "bx3"):
        dx = prev.clone().to(dev)
        if k == "fp32":
            ops.conv2d_fwd(dy.to(dev), ops.conv2d_prep_weight(w.to(dev), 1), None, dx, gd, residual=dx if acc else None)
        else:
            ops.conv1x1_bx3_fwd(dy.to(dev), ops.conv1x1_bx3_prep(w.to(dev), 1), None, dx, gd, residual=dx if acc else None)
        n += 1
        e = float((dx[:, 3:3 + Cf_in].double().cpu() - ref).abs().max() / ref.abs().max())
        e2 = float((dx[:, :3].cpu() - prev[:, :3]).abs().max()) + float((dx[:, 3 + Cf_in:].cpu() - prev[:, 3 + Cf_in:]).abs().max())
        if not (e < 5e-6 and e2 == 0):
            bad += 1
            if bad < 40:
                print("BAD %-5s N%d %dx%d fwd %d->%d acc=%d err %.2e outside %.1e" % (k, N, H, W, Cf_in, Cf_out, acc, e, e2))
print("%d cases, %d bad" % (n, bad))
