import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
dev = torch.device("cuda:0")
N = 16
def mk(C, H, W):
    x = torch.randn(N, C, H, W, device=dev); r = torch.randn(N, C, H, W, device=dev); y = torch.empty_like(x)
    g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    return dict(C=C, HW=H * W, x=x, r=r, y=y, g=g, b=b, prm=torch.empty(3, C, device=dev), gap=torch.empty(N, C, device=dev),
                dx=torch.empty_like(x), dg=torch.empty(C, device=dev), db=torch.empty(C, device=dev))
def fwd(d, gap, res):
    ops.bn_coop_fwd(d["x"], d["C"], 0, N, d["C"], d["C"], d["HW"], (d["g"], d["b"], None, None), None, 1e-5, 0.1, d["prm"], d["y"], d["C"], 0, True,
                    residual=d["r"] if res else None, r_ctot=d["C"], r_coff=0, gap_out=d["gap"] if gap else None, gap_ctot=d["C"], gap_coff=0)
def bwd(d):
    ops.bn_coop_bwd(d["r"], d["C"], 0, d["x"], d["C"], 0, d["prm"], d["b"], None, d["dx"], None, d["dg"], d["db"], None, None, False, N, d["C"], d["C"], d["HW"], True)
shapes = [(128, 64, 512), (256, 64, 256), (384, 64, 128)]
for mode in (1, 0):
    ops.bn_coop_set_mode(mode)
    for name, fn in (("fwd", lambda d: fwd(d, False, False)), ("fwd+res", lambda d: fwd(d, False, True)), ("fwd+gap", lambda d: fwd(d, True, False)),
                     ("fwd+gap+res", lambda d: fwd(d, True, True)), ("bwd", bwd), ("fwd+gap+res,bwd", lambda d: (fwd(d, True, True), bwd(d)))):
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        data = [[mk(*s) for s in shapes] for _ in streams]
        torch.cuda.synchronize()
        t0 = time.time()
        for it in range(30):
            for k in range(len(shapes)):
                for s, dd in zip(streams, data):
                    with torch.cuda.stream(s):
                        fn(dd[k])
        torch.cuda.synchronize()
        dt = time.time() - t0
        err = ops.bn_coop_errors()
        print("mode", mode, "%-18s" % name, "%.3f s" % dt, "errors", err, [e[1][:4].tolist() for e in ops._COOP_WS.values()], flush=True)
        if err:
            ops.bn_coop_check(fallback=False)
