#!/usr/bin/env python3
"""Where does conv3x3_bx3_kernel's time go?  Builds variant libraries with parts of the kernel removed
(-DBX3_ABLATE=bits, conv_bx3.hip) and times the PointSeg expand3x3 launches with each.

    python tools/bx3_ablate.py build      # here (hipcc cross-compiles): tools/micro/_abl/lib_<bits>.so
    python tools/bx3_ablate.py run        # on the GPU box: one subprocess per variant
"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ABL = os.path.join(ROOT, "tools", "micro", "_abl")
VARIANTS = [0, 1, 2, 3, 4, 8, 12, 7, 15]
NAMES = {0: "full", 1: "-split/store", 2: "-patch loads", 3: "-staging", 4: "-A loads", 8: "-B lds reads", 12: "-A -B",
         7: "-staging -A", 15: "MFMA + barrier only"}
LAYERS = [("b1.e3 fwd", 16, 64, 64, 512), ("b1.e3 dgrad", 64, 16, 64, 512), ("b2.e3 fwd", 32, 128, 64, 256),
          ("b2.e3 dgrad", 128, 32, 64, 256), ("b3.0.e3 fwd", 48, 192, 64, 128), ("b3.0.e3 dgrad", 192, 48, 64, 128),
          ("b3.2.e3 fwd", 64, 256, 64, 128), ("b3.2.e3 dgrad", 256, 64, 64, 128), ("b4.e3 fwd", 64, 256, 32, 64),
          ("b4.e3 dgrad", 256, 64, 32, 64), ("b5.e3 fwd", 80, 384, 16, 32), ("b5.e3 dgrad", 384, 80, 16, 32)]


def build():
    from deeplio_amd import build as B
    from deeplio_amd._header import abi_hash
    B.build(verbose=False)
    os.makedirs(ABL, exist_ok=True)
    objs = [os.path.join(B.OBJ, f) for f in sorted(os.listdir(B.OBJ)) if f.endswith(".o") and f != "conv_bx3.o"]
    for v in VARIANTS:
        o = os.path.join(ABL, "conv_bx3_%d.o" % v)
        subprocess.check_call([B.HIPCC] + B.FLAGS + ["-DDLIO_HEADER_CRC=%du" % abi_hash(), "-DBX3_ABLATE=%d" % v, "-c",
                               os.path.join(B.CSRC, "conv_bx3.hip"), "-o", o])
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(ABL, "lib_%d.so" % v), o] + objs)
        os.remove(o)
        print("built variant", v, flush=True)


def child():
    import torch
    from deeplio_amd import ops
    dev = torch.device("cuda:0")
    N = 16
    out = []
    for name, ci, co, H, W in LAYERS:
        x = torch.randn(N, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        y = torch.empty(N, co, H, W, device=dev)
        d = ops.conv_desc(N, ci, H, W, co, 3, 3, 1, 1, 1, 1)
        wb = ops.conv3x3_bx3_prep(w, 0)
        for _ in range(3): ops.conv3x3_bx3_fwd(x, wb, None, y, d)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): ops.conv3x3_bx3_fwd(x, wb, None, y, d)
        b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / 20 * 1e3)
    print("RESULT " + " ".join("%.1f" % t for t in out))


def run():
    lib = os.path.join(ROOT, "deeplio_amd", "libdeeplio_hip.so")
    keep = lib + ".keep"
    shutil.copy(lib, keep)
    res = {}
    try:
        for v in VARIANTS:
            shutil.copy(os.path.join(ABL, "lib_%d.so" % v), lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            if not line:
                print("variant", v, "failed:", r.stderr[-500:]); continue
            res[v] = [float(t) for t in line[0].split()[1:]]
    finally:
        shutil.move(keep, lib)
    print("us per launch, N = 16; MFMA floor = 6 x flops / 2516.8 TF/s")
    print("%-16s %7s " % ("layer", "floor") + " ".join("%9s" % NAMES[v][:9] for v in VARIANTS if v in res))
    for i, (name, ci, co, H, W) in enumerate(LAYERS):
        floor = 6 * 2.0 * 16 * H * W * ci * co * 9 / 2516.8e12 * 1e6
        print("%-16s %7.1f " % (name, floor) + " ".join("%9.1f" % res[v][i] for v in VARIANTS if v in res))
    for v in VARIANTS:
        print(v, NAMES[v])


if __name__ == "__main__":
    {"build": build, "child": child, "run": run}[sys.argv[1]]()
