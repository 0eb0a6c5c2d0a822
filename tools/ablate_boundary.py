#!/usr/bin/env python3
"""Upper bounds for the work at the step boundary (timing only, results wrong): the overlapped step with the weight re-layouts
after the optimizer (prep), the final optimizer sweep (adam), or both replaced by no-ops once the step is warm.
    python tools/ablate_boundary.py {none|prep|adam|both}"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deeplio_amd import ops, optimizer
what = sys.argv[1]
calls = [0, 0]
if what in ("prep", "both"):
    orig = ops._PrepCache._refresh_all
    def refresh(self, dev, cur):
        calls[0] += 1
        if calls[0] <= 3:
            return orig(self, dev, cur)
        for e in self.entries.values():
            w = e["ref"]()
            if w is not None and e["epoch"] >= 0:
                e.update(epoch=self.epoch, version=w._version, stream=cur.cuda_stream, event=None)
    ops._PrepCache._refresh_all = refresh
if what in ("adam", "both"):
    orig_apply = optimizer.Adam._apply
    def apply(self, lo, hi, count):
        calls[1] += 1
        if calls[1] <= 6 or lo != 0:         # the early sweep over the tail bucket stays
            return orig_apply(self, lo, hi, count)
    optimizer.Adam._apply = apply
sys.argv = [os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-isolated", "--steps", "30", "--warmup", "8"]
runpy.run_path(sys.argv[0], run_name="__main__")
