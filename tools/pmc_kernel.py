#!/usr/bin/env python3
"""per-kernel wave-state / instruction-mix summary from a rocprofv3 --pmc results db (any counter set):
   python tools/pmc_kernel.py <results.db> [name filter]"""
import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = cur.execute("select kernel_name, counter_name, sum(value), count(*), sum(end - start) from counters_collection "
                   "group by kernel_name, counter_name").fetchall()
tab = {}
for k, c, v, n, ns in rows:
    k = re.sub(r"\(anonymous namespace\)::|void ", "", k); k = re.sub(r"\(.*", "", k)
    if flt and flt not in k: continue
    tab.setdefault(k, {})[c] = v; tab[k]["_n"] = n; tab[k]["_ns"] = ns
for k, v in tab.items():
    print(k, "launches", v["_n"], "avg us %.1f" % (v["_ns"] / v["_n"] / 1e3))
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    for c, x in sorted(v.items()):
        if c.startswith("_"): continue
        print("   %-28s %14.0f  %6.1f %% of WAVE_CYCLES" % (c, x, 100.0 * x / wc))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        print("   MfmaUtil %.1f %%" % (100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)))
