#!/usr/bin/env python3
"""Upper bounds for the stem work (timing only, results wrong): the overlapped step with one stem launch family replaced by a
no-op.   python tools/ablate_stem.py {none|wgrad|bnbwd|conv|pool}"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops
what = sys.argv[1]
if what == "wgrad":
    orig = ops.conv2d_wgrad
    def f(x, dy, dw, desc, in_aff=None, accumulate=False):
        if desc.KH == 3 and desc.KW == 5:
            return dw
        return orig(x, dy, dw, desc, in_aff=in_aff, accumulate=accumulate)
    ops.conv2d_wgrad = f
elif what == "bnbwd":
    ops.bn_bwd_pool = lambda *a, **k: None
elif what == "conv":
    ops.conv3x5s2_bx3_fwd = lambda x, wt, bias, raw, d: raw
elif what == "pool":
    orig = ops.maxpool2d_fwd_aff
    cache = {}
    def g(raw, aff, *a):
        k = tuple(raw.shape)
        if k not in cache:
            cache[k] = orig(raw, aff, *a)
        return cache[k]
    ops.maxpool2d_fwd_aff = g
sys.argv = [os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-isolated", "--steps", "20", "--warmup", "5"]
runpy.run_path(sys.argv[0], run_name="__main__")
