#!/bin/bash
# round 6, job AD: cooperative BatchNorm one-item-per-workgroup mode in the step: why some runs fail; small grids of the persistent mode
cd /root/repo; mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
  DLIO_BN_COOP_MODE=1 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated > gpurun_out/r6ad_m1_$i.out 2> gpurun_out/r6ad_m1_$i.err
  echo "rc=$?" >> gpurun_out/r6ad_m1_$i.err
done
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "warm $(run X=1)" > gpurun_out/r6ad_ab.txt
for i in 1 2 3 4 5; do
  echo "default $(run X=1)"
  for c in 48 64 80 96 112; do echo "cus$c $(run DLIO_BN_COOP_CUS=$c)"; done
  echo "mode0 $(run DLIO_BN_COOP_MODE=0)"
done >> gpurun_out/r6ad_ab.txt 2>&1
python - <<'P' >> gpurun_out/r6ad_ab.txt
import collections, statistics
v = collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r6ad_ab.txt'):
    p = l.split()
    if len(p) == 2 and p[0] != 'warm':
        try: v[p[0]].append(float(p[1]))
        except ValueError: pass
for k, x in v.items():
    print("# %-10s n=%d median %.3f mean %.3f min %.3f max %.3f" % (k, len(x), statistics.median(x), statistics.mean(x), min(x), max(x)))
P
