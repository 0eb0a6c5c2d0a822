#!/usr/bin/env python3
"""print SQ counters per kernel from a rocprofv3 --pmc rocpd db"""
import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, grid_size, counter_name, avg(value), count(*) from counters_collection "
                   "group by kernel_name, grid_size, counter_name").fetchall()
tab = {}
for k, g, c, v, n in rows:
    k = re.sub(r"\(anonymous namespace\)::|void ", "", k); k = re.sub(r"\(.*", "", k)
    tab.setdefault((k, g), {})[c] = v
names = sorted({c for v in tab.values() for c in v})
for (k, g), v in sorted(tab.items()):
    if "conv" not in k and "wgrad" not in k:
        continue
    print("%-50s grid %8d" % (k[:50], g))
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    for c in names:
        if c in v:
            print("     %-28s %14.0f  (%.1f%% of WAVE_CYCLES)" % (c, v[c], 100 * v[c] / wc))
