#!/bin/bash
# round 6, job AG: one-item-per-workgroup cooperative BatchNorm with at most one such launch in flight (token): failures? speed?
cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/r6ag_*
for i in 1 2 3 4; do
  DLIO_BN_COOP_MODE=1 DLIO_CHECK_EVERY=0 WATCH_VERBOSE=1 DLIO_BN_COOP_DEBUG=1 timeout 600 python tools/step_watch.py 300 1 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r6ag_watch_$i.out
done
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated 2>gpurun_out/r6ag_last.err | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('bn_coop'))
except Exception: print('FAIL')"; }
echo "warm $(run X=1)" > gpurun_out/r6ag_ab.txt
for i in 1 2 3 4 5 6 7 8; do
  echo "default $(run X=1)"
  echo "m1token $(run DLIO_BN_COOP_MODE=1)"
  echo "m1token $(run DLIO_BN_COOP_MODE=1)"
  echo "m2token $(run DLIO_BN_COOP_TOKEN=1)"
done >> gpurun_out/r6ag_ab.txt 2>&1
python - <<'P' >> gpurun_out/r6ag_ab.txt
import collections, statistics
v = collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r6ag_ab.txt'):
    p = l.split()
    if len(p) >= 2 and p[0] != 'warm':
        try: v[p[0]].append(float(p[1]))
        except ValueError: v[p[0] + '_FAIL'].append(0.0)
for k, x in v.items():
    print("# %-10s n=%d median %.3f mean %.3f min %.3f max %.3f" % (k, len(x), statistics.median(x), statistics.mean(x), min(x), max(x)))
P
