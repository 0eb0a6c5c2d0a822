#!/bin/bash
# round 6, job Z: what the step boundary (final optimizer sweep + weight re-layouts) costs: ablations, 5 alternations
cd /root/repo; mkdir -p gpurun_out
run() { python tools/ablate_boundary.py $1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "warm $(run none)" > gpurun_out/r6z_ab.txt
for i in 1 2 3 4 5; do
  for w in none prep adam both; do echo "$w $(run $w)"; done
done >> gpurun_out/r6z_ab.txt 2>&1
python - <<'P' >> gpurun_out/r6z_ab.txt
import collections, statistics
v = collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r6z_ab.txt'):
    p = l.split()
    if len(p) == 2 and p[0] != 'warm':
        try: v[p[0]].append(float(p[1]))
        except ValueError: pass
for k, x in v.items():
    print("# %-10s n=%d median %.3f mean %.3f min %.3f max %.3f" % (k, len(x), statistics.median(x), statistics.mean(x), min(x), max(x)))
P
