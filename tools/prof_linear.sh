cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_lin -o lin -- python $R/tools/bench_linear.py > $R/gpurun_out/prof_lin.log 2>&1
DB=$(find $R/gpurun_out/prof_lin -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/lin_kernel_stats.md 1 "bench_linear"
python - <<PY
import sqlite3
cur = sqlite3.connect("$DB").cursor()
# per kernel name and grid size: avg
for r in cur.execute("select name, grid_size_x, grid_size_y, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels group by name, grid_size_x, grid_size_y order by name").fetchall():
    print("%-70s grid %6d x %4d  n=%4d avg %7.2f us min %7.2f" % (r[0][:70], r[1], r[2], r[3], r[4], r[5]))
PY
rm -rf $R/gpurun_out/prof_lin
