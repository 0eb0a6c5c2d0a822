#!/usr/bin/env python3
"""What does a kernel family cost the OVERLAPPED step?  Replaces the named deeplio_amd.ops entry points by no-ops
(results are wrong: timing only) and runs bench.py in-process.
   python tools/ablate_family.py conv2d_wgrad [more ops...] -- [bench flags]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
names = sys.argv[1:sys.argv.index("--")] if "--" in sys.argv else sys.argv[1:]
rest = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
RET = {"conv2d_wgrad": 2, "bn_bwd_fused": 8, "bn_train_apply": None, "maxpool2d_bwd": None}
for n in names:
    idx = RET.get(n, None)
    if n == "conv2d_wgrad":
        setattr(ops, n, lambda x, dy, dw, desc, in_aff=None, accumulate=False: dw)
    elif n == "bn_bwd_fused":
        setattr(ops, n, lambda *a, **k: a[8])
    else:
        raise SystemExit("no no-op known for " + n)
sys.argv = [os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-isolated"] + rest
runpy.run_path(sys.argv[0], run_name="__main__")
