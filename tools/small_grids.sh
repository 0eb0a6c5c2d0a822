cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_g -o g -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-isolated --serial > $R/gpurun_out/prof_g.log 2>&1
DB=$(find $R/gpurun_out/prof_g -name '*.db' | head -1)
python - <<PY
import sqlite3, re
cur = sqlite3.connect("$DB").cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
print(cols)
gx = [c for c in cols if 'grid' in c.lower()]; wx = [c for c in cols if 'workgroup' in c.lower() or 'block' in c.lower()]
print(gx, wx)
q = "select name, %s, %s, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels group by name, %s, %s order by 6 desc" % (gx[0], wx[0], gx[0], wx[0])
rows = cur.execute(q).fetchall()
print("few-workgroup launches (< 512 workgroups) by total time:")
n = 0
for name, g, w, c, avg, tot in rows:
    wg = g // max(w, 1)
    if wg < 512 and avg > 8:
        name = re.sub(r"\(anonymous namespace\)::|void ", "", name); name = re.sub(r"\(.*", "", name)[:60]
        print("%-60s wgs %5d x %4d thr  n=%4d avg %7.1f us  total %7.2f ms" % (name, wg, w, c, avg, tot))
        n += 1
        if n > 40: break
PY
rm -rf $R/gpurun_out/prof_g
