#!/usr/bin/env python3
"""Per-layer efficiency of the conv kernels from a rocprofv3 rocpd db of bench.py (PointSeg headline config)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
steps = int(sys.argv[2]); N = 16
layers = []; H, W = 64, 512
blocks = [("b1", [(64,16,64),(128,16,64)], (1,2)), ("b2", [(128,32,128),(256,32,128)], (1,2)),
          ("b3", [(256,48,192),(384,48,192),(384,64,256),(512,64,256)], (2,2)), ("b4", [(512,64,256),(512,64,256)], (2,2)),
          ("b5", [(512,80,384),(768,80,384)], None)]
for bn, fires, pool in blocks:
    for i, (ci, sq, e) in enumerate(fires):
        layers += [(f"{bn}.{i}.sq", ci, sq, 1, H, W), (f"{bn}.{i}.e1", sq, e, 1, H, W), (f"{bn}.{i}.e3", sq, e, 3, H, W)]
    if pool: H, W = H // pool[0], W // pool[1]
cd = lambda a, b: (a + b - 1) // b
exp = {}
for nm, ci, co, k, H, W in layers:
    fl = 2.0 * N * H * W * ci * co * k * k
    byt = 4.0 * N * H * W * (ci + co)
    for dname, (a, b) in (("fwd", (ci, co)), ("dgrad", (co, ci))):
        if k == 1:
            mr = 1 if b <= 32 else (3 if 64 < b <= 96 else 2)
            key = ("1x1", mr, cd(H * W, 256) * cd(b, 32 * mr) * N * 256)
        else:
            twn = 2 if W > 32 else 1
            key = ("3x3", twn, cd(W, 32 * twn) * cd(H, 4) * cd(b, 64) * N * 256)
        exp.setdefault(key, []).append((nm, dname, fl, byt))
    # wgrad
    nt = 2 if k == 1 else 5
    ck = nt * 32 // (k * k)
    pairs = cd(co, 64) * cd(ci, ck)
    tiles = N * cd(W, 32) * cd(H, 4)
    splits = min(cd(512, pairs), tiles)
    slab = co * ci * k * k * 4
    if splits * slab > (96 << 20): splits = (96 << 20) // slab
    exp.setdefault(("wg%d" % k, 0, pairs * splits * 256), []).append((nm, "wgrad", fl, byt))
rows = cur.execute("select name, grid_x, count(*), avg(end-start)/1e3 from kernels where name like '%conv%kernel%' group by name, grid_x").fetchall()
out = []
for name, grid, c, avg in rows:
    if "conv1x1_direct" in name:
        key = ("1x1", int(re.search(r"<(\d)", name).group(1)), grid)
    elif "conv_fwd_kernel<3, 3, 1, 1" in name:
        key = ("3x3", int(re.search(r", (\d)>", name).group(1)), grid)
    elif "conv_wgrad_kernel<3, 3, 1, 1" in name: key = ("wg3", 0, grid)
    elif "conv_wgrad_kernel<1, 1, 1, 1" in name: key = ("wg1", 0, grid)
    else: continue
    cands = exp.get(key, [])
    if not cands: out.append((avg * c / steps / 1e3, key, grid // 256, c / steps, avg, 0, 0, "?")); continue
    fl = sum(x[2] for x in cands) / len(cands); by = sum(x[3] for x in cands) / len(cands)
    out.append((avg * c / steps / 1e3, key, grid // 256, c / steps, avg, fl / (avg * 1e-6) / 1e12, by / (avg * 1e-6) / 1e12,
                ",".join("%s:%s" % (x[0], x[1]) for x in cands[:4])))
out.sort(reverse=True)
print("ms/step kernel blocks n/step avg_us TF/s TB/s layers")
for o in out[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%6.2f %-12s %6d %4.1f %7.1f %6.1f %5.2f  %s" % (o[0], "%s/%d" % (o[1][0], o[1][1]), o[2], o[3], o[4], o[5], o[6], o[7]))
