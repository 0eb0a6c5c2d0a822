import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
from deeplio_amd import functional as Fh
dev = torch.device("cuda:0")
N = 8
LAYERS = [("conv2", 64, 128, 64, 1024, 3, 5, 1, 2, 1, 2), ("conv3", 128, 256, 64, 512, 3, 5, 1, 2, 1, 2), ("conv4", 256, 512, 64, 256, 3, 3, 2, 2, 1, 1),
          ("conv5", 512, 512, 32, 128, 3, 3, 2, 2, 1, 1), ("conv6", 512, 1024, 16, 64, 3, 3, 2, 2, 1, 1)]
for name, Cin, Cout, H, W, KH, KW, SH, SW, PH, PW in LAYERS:
    x = torch.randn(N, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, KH, KW, device=dev) / math.sqrt(Cin * KH * KW)
    d = ops.conv_desc(N, Cin, H, W, Cout, KH, KW, SH, SW, PH, PW)
    y, y3 = torch.empty(N, Cout, d.OH, d.OW, device=dev), torch.empty(N, Cout, d.OH, d.OW, device=dev)
    am = x.abs().max().reshape(1).contiguous()
    ops.conv_h2_strided_fwd(x, am, ops.conv_h2_prepped(w, 0), None, y, d)
    torch.cuda.synchronize(); print(name, "fwd h2 ok", flush=True)
    ops.conv3x5s2_bx3_fwd(x, ops.conv_bx3_prepped(w, 0), None, y3, d)
    torch.cuda.synchronize()
    print(name, "fwd err vs bx3", float((y - y3).abs().max() / y3.abs().max()), flush=True)
    # phases
    dy = torch.randn(N, Cout, d.OH, d.OW, device=dev) * 1e-4
    plan = Fh._phase_plan(d)
    amy = dy.abs().max().reshape(1).contiguous()
    for it in plan:
        if it is None: continue
        rh, rw, Mh, Mw, pt, pl, Hp, Wp = it
        if not Fh._phase_on_bx3(d, Mh, Mw): continue
        g = ops.conv_desc(N, Cout, d.OH, d.OW, Cin, Mh, Mw, 1, 1, pt, pl, OH=Hp, OW=Wp, in_ctot=Cout, in_coff=0, out_ctot=Cin, out_coff=0)
        o1, o2 = torch.empty(N, Cin, Hp, Wp, device=dev), torch.empty(N, Cin, Hp, Wp, device=dev)
        ops.conv_h2_taps_fwd(dy, amy, ops.conv_h2_prepped_phase(w, SH, SW, rh, rw, cache=False), None, o1, g)
        torch.cuda.synchronize(); print(name, "phase", (rh, rw, Mh, Mw), "h2 ok", flush=True)
        ops.conv_bx3_taps_fwd(dy, ops.conv_bx3_prepped_phase(w, SH, SW, rh, rw, cache=False), None, o2, g)
        torch.cuda.synchronize()
        print(name, "phase err", float((o1 - o2).abs().max() / o2.abs().max()), flush=True)
