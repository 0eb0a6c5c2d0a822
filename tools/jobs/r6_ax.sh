#!/bin/bash
# round 6, job AX: grid cap of the persistent cooperative launches that remain under mode 3 (fire_blk1), 5 alternations
cd /root/repo; mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "warm $(run X=1)" > gpurun_out/r6ax_ab.txt
for i in 1 2 3 4 5; do
  echo "cus160 $(run X=1)"
  for c in 96 128 208 256; do echo "cus$c $(run DLIO_BN_COOP_CUS=$c)"; done
done >> gpurun_out/r6ax_ab.txt 2>&1
python - <<'P' >> gpurun_out/r6ax_ab.txt
import collections, statistics
v = collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r6ax_ab.txt'):
    p = l.split()
    if len(p) == 2 and p[0] != 'warm':
        try: v[p[0]].append(float(p[1]))
        except ValueError: pass
for k, x in v.items():
    print("# %-8s n=%d median %.3f mean %.3f min %.3f max %.3f" % (k, len(x), statistics.median(x), statistics.mean(x), min(x), max(x)))
P
