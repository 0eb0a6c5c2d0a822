#!/bin/bash
# round 6, job V: exact scale vs analytic bound: ten alternations + exclusive kernel times
cd /root/repo; mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "warm $(run X=1)" > gpurun_out/r6v_ab.txt
for i in 1 2 3 4 5 6 7 8 9 10; do
  echo "exact $(run X=1)"
  echo "bound $(run DLIO_SPLIT16_EXACT=0)"
done >> gpurun_out/r6v_ab.txt 2>&1
python - <<'P' >> gpurun_out/r6v_ab.txt
import collections, statistics
v = collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r6v_ab.txt'):
    p = l.split()
    if len(p) == 2 and p[0] != 'warm':
        try: v[p[0]].append(float(p[1]))
        except ValueError: pass
for k, x in v.items():
    print("# %-10s n=%d median %.3f mean %.3f min %.3f max %.3f" % (k, len(x), statistics.median(x), statistics.mean(x), min(x), max(x)))
P
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
DLIO_SPLIT16_EXACT=$v rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_v$v -o pv -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-isolated --serial > /root/repo/gpurun_out/prof_v.log 2>&1
DB=$(find /root/repo/gpurun_out/prof_v$v -name '*.db' | head -1)
python /root/repo/tools/rocprof_summary.py $DB /root/repo/gpurun_out/r6v_kernel_stats_serial_exact$v.md 8 "bench.py --serial DLIO_SPLIT16_EXACT=$v"
rm -rf /root/repo/gpurun_out/prof_v$v
done
