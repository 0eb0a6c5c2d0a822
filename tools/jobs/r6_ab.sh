#!/bin/bash
# round 6, job AB: vectorised batched weight re-layouts: tests, alternations against the previous kernels, exclusive times
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "two_piece or h2 or magnitude or prep or relayout or bx3 or split" 2>&1 | tail -3 ) > gpurun_out/r6ab_t.log
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
runold() { python tools/variant_lib.py run prepold -- env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "warm $(run X=1)" > gpurun_out/r6ab_ab.txt
for i in 1 2 3 4 5 6 7 8; do
  echo "vec $(run X=1)"
  echo "scalar $(runold X=1)"
done >> gpurun_out/r6ab_ab.txt 2>&1
python - <<'P' >> gpurun_out/r6ab_ab.txt
import collections, statistics
v = collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r6ab_ab.txt'):
    p = l.split()
    if len(p) == 2 and p[0] != 'warm':
        try: v[p[0]].append(float(p[1]))
        except ValueError: pass
for k, x in v.items():
    print("# %-10s n=%d median %.3f mean %.3f min %.3f max %.3f" % (k, len(x), statistics.median(x), statistics.mean(x), min(x), max(x)))
P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_ab -o py -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-isolated --serial > /root/repo/gpurun_out/prof_ab.log 2>&1
DB=$(find /root/repo/gpurun_out/prof_ab -name '*.db' | head -1)
python /root/repo/tools/rocprof_summary.py $DB /root/repo/gpurun_out/r6ab_kernel_stats_serial.md 8 "bench.py --serial"
rm -rf /root/repo/gpurun_out/prof_ab
